"""Print the per-dispatch timeline of the last bench step from a rocprofv3 results.db (kernel trace)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
                        f"from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
def short(nm):
    m = re.search(r'(k_\w+?)I', nm)
    if m:
        args = re.findall(r'Li(\d+)E|Lb([01])E', nm)
        return m.group(1) + "<" + ",".join(a or b for a, b in args) + ">"
    return nm[:48]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 40
last = rows[-nlast:]
t0 = last[0][1]
prev_end = None
for nm, s, e, gx, gy, gz, wx in last:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f}  +{gap:5.1f}  {(e - s) / 1e3:8.1f} us  grid {gx // max(wx, 1)}x{gy}x{gz}  {short(nm)}")
    prev_end = e
