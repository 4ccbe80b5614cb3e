"""Per-kernel averages from a rocprofv3 rocpd database: python tools/kstats.py file.db [substr]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
q = """select s.kernel_name, k.grid_size_x, k.grid_size_y, count(*), avg(k.end-k.start)/1000.0, min(k.end-k.start)/1000.0
       from rocpd_kernel_dispatch k join rocpd_info_kernel_symbol s on k.kernel_id=s.id
       group by s.kernel_name, k.grid_size_x, k.grid_size_y order by 5 desc"""
for name, gx, gy, cnt, avg, mn in con.execute(q):
    if sub in name:
        print("%-100s grid=(%d,%d) n=%d avg=%.1f us min=%.1f us" % (name[:100], gx, gy, cnt, avg, mn))
