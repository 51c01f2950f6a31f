"""diagnostic: which librccl does dlopen("librccl.so.1") give a process that has imported torch, and does ncclCommInitRank work there?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else 'torch_first'
if mode == 'torch_first':
    import torch
    import torch.multiprocessing
from xcontour_amd import _native as nat
ctx = nat.Context(0)
if mode == 'torch_after_ctx':
    import torch
    import torch.multiprocessing
if len(sys.argv) > 2:
    print('info before', ctx.comm_info())
uid = ctx.comm_unique_id()
try:
    c = ctx.comm_create(1, 0, uid)
    ctx.comm_attach(c, 1, 0)
    print(mode, 'comm ok', ctx.comm_info())
except Exception as e:
    print(mode, 'comm FAILED', e, ctx.comm_info())
maps = [l.split()[-1] for l in open('/proc/self/maps') if 'rccl' in l or 'libamdhip64' in l or 'libhsa-runtime' in l]
print(sorted(set(maps)))
