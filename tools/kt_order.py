"""Kernel durations of a rocprofv3 results database in launch order: python tools/kt_order.py <db> [name filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for r in c.execute("select substr(name,1,64), grid_x, grid_z, (end-start)/1000.0 from kernels where name like ? order by start", ('%' + flt + '%',)):
    print("%-64s grid %7d z %4d  %8.1f us" % r)
