#!/usr/bin/env python
"""A/B of whole-step time between environment settings (dev): one subprocess per setting, rounds alternate.
python tools/dev/env_ab.py [--b 16] [--rounds 2] name:VAR=val[,VAR=val] ...      ('name:' alone = the default environment)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
a = sys.argv[1:]
B, rounds, sets = 16, 2, []
i = 0
while i < len(a):
    if a[i] == '--b':
        B = int(a[i + 1]); i += 2
    elif a[i] == '--rounds':
        rounds = int(a[i + 1]); i += 2
    else:
        n, _, kv = a[i].partition(':')
        sets.append((n, dict(x.split('=') for x in kv.split(',') if x))); i += 1
for r in range(rounds):
    for n, kv in sets:
        env = dict(os.environ)
        env.update(kv)
        res = subprocess.run([sys.executable, os.path.join(HERE, 'lib_ab.py'), '--child', str(B), '3'], env=env, capture_output=True, text=True)
        line = [l for l in res.stdout.splitlines() if l.startswith('AB')]
        print(f'round {r} {n:12s}', line[0] if line else ('FAILED ' + res.stderr[-600:]), flush=True)
