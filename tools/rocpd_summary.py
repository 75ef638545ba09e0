"""Per-kernel summary (count / total / average / share) from a rocprofv3 rocpd SQLite database
(the default output of `rocprofv3 --kernel-trace --stats`)."""
import re
import sqlite3
import sys


def summarize(path):
    con = sqlite3.connect(path)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = con.execute(
        'select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), '
        'max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name '
        'order by 3 desc' % (disp, sym)).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ['%-100s %8s %12s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_us', 'avg_us',
                                                      'min_us', 'max_us', 'share')]
    for name, n, tot, mn, mx in rows:
        short = re.sub(r'\(.*$', '', name)[:100]
        lines.append('%-100s %8d %12.1f %12.2f %12.2f %12.2f %6.2f%%' %
                     (short, n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    lines.append('total kernel time: %.3f ms' % (total / 1e6))
    return '\n'.join(lines)


if __name__ == '__main__':
    print(summarize(sys.argv[1]))
