#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S) for loops whose every trip waits for its own memory loads: loops with global/flat
loads and an `s_waitcnt vmcnt(0)` inside -- each trip then costs a full memory round trip.
Usage: python tools/scan_load_loops.py file.s [file.s ...]"""
import re
import sys

for f in sys.argv[1:]:
    txt = open(f).read()
    for name in re.findall(r'^(_Z\w+):', txt, re.M):
        i = txt.index('\n' + name + ':')
        j = txt.index('s_endpgm', i)
        cur, stats = None, {}
        for l in txt[i:j].split('\n'):
            m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
            if m:
                cur = m.group(1) if 'Loop' in m.group(2) else None
            if cur:
                st = stats.setdefault(cur, [0, 0, 0])
                if re.match(r'^\t(global|flat|buffer|scratch)_load', l):
                    st[0] += 1
                if re.search(r's_waitcnt vmcnt\(0\)', l):
                    st[1] += 1
                st[2] += 1
        hot = {k: v for k, v in stats.items() if v[0] > 0 and v[1] > 0}
        if hot:
            print(name[:44], ' '.join('%s[loads %d waits %d len %d]' % (k, *v) for k, v in hot.items()))
