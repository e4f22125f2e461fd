"""Per-launch durations of one fused rollout from a rocprofv3 kernel trace: python tools/yield_trace.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[-int(sys.argv[2]) if len(sys.argv) > 2 else 0:]:
  name = r['Kernel_Name'].split('(')[0].replace('void blcd::', '')[:40]
  print(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} us  +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us  grid {r.get('Grid_Size','?'):>8}  {name}")
