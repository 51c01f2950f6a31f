#!/usr/bin/env python3
"""Runs the python block of README.md verbatim (needs the GPU)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
code = re.search(r"```python\n(.*?)```", open('README.md').read(), re.S).group(1)
ns = {}
exec(code, ns)
print('README example ok: LWA max %.3f, Q %r, crossing %r' % (float(ns['lwa'].values.max()), ns['Qx'].shape, [b.shape for b in ns['bc']]))
