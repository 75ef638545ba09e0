"""VERDICT r3 item 4 (one launch for the tiles of a step), priced before building it.

What could a launch that carries all four tiles of a `--size 1448 --tile-size 1024` step (4 x 724^2)
gain over four tile evaluations on four HIP streams?  A batched launch fills whole rounds of 256
workgroups "by construction"; its time per layer is the single-tile time scaled by
ceil(4 n / 256) / (4 ceil(n / 256)) (n = workgroups of the layer for one tile, K slices included).
Summed over the convolution layers of profiles/r03_per_layer_times.txt (lone 724^2 tile, HIP events)
that is the BOUND a batched launch of the same kernels cannot beat.

    python tools/packing_bound.py

Result (also in profiles/r04_packing_bound_724.txt): 15.99 ms per step for the tiles + ~0.3 ms of
image ops; the four streams already run the step in 15.5 ms (profiles/r03_scale_steps.txt, 15.4-15.6
in round 4).  The workgroups of the four streams' kernels interleave on the CUs as they become
free, which packs at least as well as whole rounds do -- the tile index as a grid dimension of
every kernel would rewrite ~20 kernel signatures for nothing on this workload.  What is left at
724^2 is inside the kernels: 6 % patch padding per axis (724 -> 768, 181 -> 192, 91 -> 96) and
the per-workgroup prologue / epilogue, not the packing.
"""
import math

# (name, K, M, plane side, ms forward, ms backward) of a lone 724^2 tile
LAYERS = [('conv1_2', 64, 64, 724, .194, .198), ('conv2_1', 64, 128, 362, .102, .106),
          ('conv2_2', 128, 128, 362, .184, .182), ('conv3_1', 128, 256, 181, .110, .119),
          ('conv3_2', 256, 256, 181, .198, .215), ('conv3_3', 256, 256, 181, .198, .200),
          ('conv3_4', 256, 256, 181, .198, .200), ('conv4_1', 256, 512, 91, .125, .106),
          ('conv4_2', 512, 512, 91, .206, .225), ('conv4_3', 512, 512, 91, .206, .228),
          ('conv4_4', 512, 512, 91, .205, .205), ('conv5_1', 512, 512, 46, .059, .058)]
TILE_MS = 4.657          # all kernel groups of the lone tile


def geometry(H):         # conv_wino2.hip: wino2_pick_geometry
    if H <= 40:
        return 16, 16
    wide = math.ceil(H / 4) * math.ceil(H / 64)
    mid = math.ceil(H / 8) * math.ceil(H / 32)
    return (8, 32) if mid * 10 <= wide * 9 else (4, 64)


def ksplit(K, n, M, H):  # conv_wino2.hip: wino2_splitk_factor
    chunks, out_mb, best = math.ceil(K / 8), 4e-6 * M * H * H, None
    for f in range(1, 9):
        if chunks // f < 4:
            break
        cost = math.ceil(n * f / 256) * (chunks / f * 2.05 + 6)
        if f > 1:
            cost += (f + 1) * out_mb / 3 + 5
        if best is None or cost < best[0]:
            best = (cost, f)
    return best[1]


alone = packed = 0.0
for name, K, M, H, tf, tb in LAYERS:
    pr, pc = geometry(H)
    for k, m, t, d in ((K, M, tf, 'fwd'), (M, K, tb, 'bwd')):
        n = math.ceil(m / 64) * math.ceil(H / pr) * math.ceil(H / pc)
        f = ksplit(k, n, m, H)
        ra, r4 = math.ceil(n * f / 256), math.ceil(4 * n * f / 256) / 4
        alone += t
        packed += t * r4 / ra
        print('%s %-8s %3d->%3d @%3d  patch %2dx%-2d  %5d workgroups x %d K slices  rounds: alone %2d, '
              'with three more tiles %5.2f per tile  %.3f -> %.3f ms' % (d, name, k, m, H, pr, pc, n, f, ra, r4, t, t * r4 / ra))
other = TILE_MS - alone
print('convolutions of a lone tile %.3f ms; packed in whole rounds together with three more tiles %.3f ms '
      'per tile; other kernels %.3f ms' % (alone, packed, other))
print('bound of a batched launch for 4 x 724^2: %.2f ms per step (+ ~0.3 ms of image ops); '
      'four streams, measured: 15.5 ms per step' % (4 * (packed + other)))
