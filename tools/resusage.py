"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin or file)."""
import re, subprocess, sys
cur = None; rows = {}
for l in open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-R', l)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(['c++filt'] + list(rows), capture_output=True, text=True).stdout.split('\n')
for n, (k, v) in zip(names, rows.items()):
    print(f"{n[:62]:62s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} scratch={v.get('ScratchSize')} spill={v.get('VGPRs Spill')} occ={v.get('Occupancy')} sgpr={v.get('TotalSGPRs')}")
