"""List the dispatches of kernels whose name contains argv[2], in time order: python tools/kt_list.py <db> <substr>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, s, e, gx in c.execute("select name, start, end, grid_x from kernels where name like ? order by start", ("%" + sys.argv[2] + "%",)):
    print("%8.1f us  grid %6d  %s" % ((e - s) / 1e3, gx, name[:60]))
