"""Flags runs of global loads that hipcc serialised (a full `s_waitcnt vmcnt(0)` between consecutive loads with no
other vector-memory work between them) in a gfx950 ISA listing: `hipcc -S --cuda-device-only` output.
usage: python tools/isa_serial_loads.py file.s [min_run]"""
import re, sys
path = sys.argv[1]; min_run = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fn = None; runs = {}
state_loads_since_wait = 0; run = 0; run_start = 0; last_was_wait0 = False
for ln, line in enumerate(open(path), 1):
    s = line.strip()
    m = re.match(r'^(_Z\w+):', s)
    if m:
        fn = m.group(1); run = 0; state_loads_since_wait = 0; continue
    if not s or s.startswith(('.', ';', '//')) or fn is None: continue
    op = s.split()[0]
    if op.startswith(('global_load', 'buffer_load', 'flat_load')):
        state_loads_since_wait += 1
    elif op == 's_waitcnt' and ('vmcnt(0)' in s):
        if state_loads_since_wait == 1:
            if run == 0: run_start = ln
            run += 1
        else:
            if run >= min_run: runs.setdefault(fn, []).append((run_start, run))
            run = 0
        state_loads_since_wait = 0
    elif op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_barrier')):
        pass
if run >= min_run and fn: runs.setdefault(fn, []).append((run_start, run))
for f, rs in runs.items():
    print(f[:110], ' '.join('L%d x%d' % r for r in rs))
