"""The command-line driver end to end on the GPU, from model FILES: a deploy prototxt and a
binary .caffemodel are written, read back by the product's own readers (netspec.parse_prototxt,
weights.read_caffemodel) and drive BASELINE config 4 in miniature -- VGG-19 with MAX pooling,
-o lbfgs, two scales, the second with the ragged 3 x 3 tiling of style_transfer.py:619-632 --
whose reference run (the reference's own transfer_multiscale, Progress, StatLogger and
get_image_comment, tests/golden/make_golden.py section 4c) supplies every expected value:
per-step losses, the console step line (style_transfer.py:950-951), the <RUN>_log.csv columns
(style_transfer.py:121-130), the --save-every files (style_transfer.py:938-939), the PNG iTXt
comment (style_transfer.py:1003-1010) and the final picture."""

import csv
import glob
import os
import re

import numpy as np
import pytest
from PIL import Image

from style_transfer_amd import cli, netspec, weights

pytestmark = pytest.mark.gpu

STEP_LINE = re.compile(r'^Step (\d+), time: (\d+\.\d\d) s, update: (\d+\.\d\d), '
                       r'loss: (\d\.\d{6}e[+-]\d\d), tv: (\d+\.\d\d)$')


def _steps(text):
    rows = []
    for line in text.splitlines():
        if line.startswith('Step '):
            m = STEP_LINE.match(line)
            assert m, 'step line does not have the reference format: %r' % line
            rows.append([float(v) for v in m.groups()])
    return np.array(rows)


@pytest.mark.parametrize('kernels', ['fp32', 'default'])
def test_cli_config4_miniature_from_model_files(golden, tmp_path, monkeypatch, capsys, kernels):
    """kernels 'fp32': the fp32-MFMA kernels only -- held to the COMMITTED run alone; 'default': the shipped
    fp16-split kernels -- one of the reference's own branches (tests/golden/branch_sets.py cfg4)."""
    if kernels == 'fp32':
        from tests.gpu_helpers import FP32_KERNELS
        for k, v in FP32_KERNELS.items():
            monkeypatch.setenv(k, v)
    net = netspec.builtin_net('vgg19')
    proto, model = tmp_path / 'deploy.prototxt', tmp_path / 'w.caffemodel'
    proto.write_text(netspec.to_prototxt(net))
    weights.write_caffemodel(str(model), weights.synthetic_weights(net, 0))
    Image.fromarray(golden['e2e_cfg4.content_u8']).save(tmp_path / 'c.png')
    Image.fromarray(golden['e2e_cfg4.style_u8']).save(tmp_path / 's.png')
    monkeypatch.chdir(tmp_path)
    argv = str(golden['e2e_cfg4.argv']).split() + ['--model', str(proto), '--weights', str(model),
                                                   '--devices', '0']
    assert cli.main(argv) == 0
    out = capsys.readouterr().out

    # ---- console: same step-line format, same step numbers, losses to 2e-4
    ref_steps, got_steps = _steps(str(golden['e2e_cfg4.step_lines'])), _steps(out)
    assert got_steps.shape == ref_steps.shape == (5, 5)
    assert np.array_equal(got_steps[:, 0], ref_steps[:, 0])
    # the reference's own trajectory on this fixture branches under another float32 implementation of its
    # convolutions (tests/helpers.cfg4_reference_branches, tests/golden/branch_sets.py): the shipped kernels
    # follow one of the reference's branches, the fp32-MFMA kernels the committed one -- to the fixture's 2e-4
    from tests.helpers import cfg4_reference_branches, matching_branch
    branches = cfg4_reference_branches(golden)
    br = matching_branch(branches[:1] if kernels == 'fp32' else branches, got_steps[:, 3])
    assert br is not None, (got_steps[:, 3], [b['log'][:, 2] for b in branches])
    ref_log = br['log']
    assert np.allclose(got_steps[:, 2], ref_log[:, 1], atol=0.025)            # update size (two decimals)
    assert np.allclose(got_steps[:, 4], ref_log[:, 3], atol=0.025)            # tv statistic
    # the 3 x 3 tiling of the 92 x 100 scale (30/30/32 x 33/33/34) was really used
    assert 'Using 3x3 tiles of size 33x30' in out
    assert re.search(r'Run \d{6}_\d{6} ending after \d+m \d+\.\d{3}s\.', out)     # wall-clock line

    # ---- <RUN>_log.csv: the reference's header, its integer columns exactly, its losses
    logs = glob.glob(str(tmp_path / '*_log.csv'))
    assert len(logs) == 1
    with open(logs[0], newline='') as f:
        rows = list(csv.reader(f))
    assert ','.join(rows[0]) == str(golden['e2e_cfg4.csv_header'])
    ref_rows = [r.split(',') for r in str(golden['e2e_cfg4.csv_rows']).splitlines()]
    assert len(rows) - 1 == len(ref_rows)
    for got, ref in zip(rows[1:], ref_rows):
        for col in (0, 1, 2, 4, 5):            # iteration, scale, step, content_h, content_w
            assert got[col] == ref[col], (got, ref)
        assert float(got[3]) >= 0                                                  # time
    assert np.allclose([float(r[7]) for r in rows[1:]], ref_log[:, 2], rtol=2e-4)  # loss

    # ---- --save-every 2: intermediate pictures under the reference's names
    run = os.path.basename(logs[0])[:-len('_log.csv')]
    saved = sorted(os.path.basename(p)[len(run):] for p in glob.glob(str(tmp_path / (run + '_out*'))))
    assert saved == sorted(str(golden['e2e_cfg4.saved_files']).split() + ['_out.png'])

    # ---- final picture + iTXt comment
    final = Image.open(tmp_path / (run + '_out.png'))
    got_u8 = np.asarray(final.convert('RGB')).astype(int)
    diff = np.abs(got_u8 - br['final_u8'].astype(int))
    assert diff.max() <= 2 and diff.mean() < 0.05, (diff.max(), diff.mean())
    comment = final.text['Comment'].splitlines()
    ref_comment = str(golden['e2e_cfg4.image_comment']).splitlines()
    assert len(comment) == len(ref_comment)
    assert comment[2].startswith('Command line: style_transfer.py -ci c.png -si s.png --size 100')
    assert comment[4] == ref_comment[4] == 'Parameters:'
    names = lambda line: re.findall(r'(\w+)=', line)
    assert comment[5].startswith('ns: Namespace(') and names(comment[5]) == names(ref_comment[5])
    assert comment[6] == ref_comment[6]        # state_obj: Namespace(scale=1, step=1, steps=2, ...)
