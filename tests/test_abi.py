"""The C-ABI library builds, loads and exports every symbol include/xcontour_hip.h declares.
No compute calls here (no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'xcontour_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(xc_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from xcontour_amd import _native as nat
    assert os.path.exists(nat.LIB_PATH), 'build it: python -c "import __graft_entry__ as g; g.build()"'
    lib = C.CDLL(nat.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), 'missing export %s' % s
    # and the binding covers exactly the header
    assert sorted(nat.PROTOTYPES) == syms


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of the two descriptor structs as gcc sees them in include/xcontour_hip.h == the ctypes mirrors"""
    import subprocess
    from xcontour_amd import _native as nat
    fields = {'xc_hist_desc': [f[0] for f in nat.HistDesc._fields_], 'xc_keff_desc': [f[0] for f in nat.KeffDesc._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % os.path.join(ROOT, 'include', 'xcontour_hip.h'), 'int main(void){']
    for st, fl in fields.items():
        src.append('printf("%s %%zu\\n", sizeof(struct %s));' % (st, st))
        for f in fl:
            src.append('printf("%s.%s %%zu\\n", offsetof(struct %s, %s));' % (st, f, st, f))
    src.append('return 0;}')
    c = tmp_path / 'layout.c'
    c.write_text('\n'.join(src))
    exe = str(tmp_path / 'layout')
    subprocess.run(['gcc', '-std=c99', '-o', exe, str(c)], check=True)
    got = dict(line.split() for line in subprocess.run([exe], check=True, stdout=subprocess.PIPE, universal_newlines=True).stdout.splitlines())
    for st, cls in (('xc_hist_desc', nat.HistDesc), ('xc_keff_desc', nat.KeffDesc)):
        assert int(got[st]) == C.sizeof(cls), st
        for f in fields[st]:
            assert int(got['%s.%s' % (st, f)]) == getattr(cls, f).offset, '%s.%s' % (st, f)
    lib = nat.load()
    assert lib.xc_version().startswith(b'xcontour_hip')


def test_integration_stub_struct_matches_the_library():
    """the ctypes struct INTEGRATION.md shows a maintainer of the reference (the `_HistDesc` of the stub) has the fields, order and
    size of the binding's own mirror of xc_hist_desc -- a stub that lags the header would let the library read past its end"""
    from xcontour_amd import _native as nat
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    body = text[text.index('class _HistDesc(C.Structure):'):]
    body = body[body.index('_fields_ = ['):body.index(']', body.index("('reserved0'")) + 1]
    ns = {'C': C, '_vp': C.c_void_p, '_i32': C.c_int32, '_i64': C.c_int64}
    exec(body.replace('_fields_', 'fields'), ns)
    stub = type('Stub', (C.Structure,), {'_fields_': ns['fields']})
    assert [f[0] for f in ns['fields']] == [f[0] for f in nat.HistDesc._fields_]
    assert C.sizeof(stub) == C.sizeof(nat.HistDesc)
    for name, _ in [(f[0], f[1]) for f in ns['fields']]:
        assert getattr(stub, name).offset == getattr(nat.HistDesc, name).offset, name


def test_no_cpu_fallback_without_device():
    """Without a GPU the product path must fail loudly, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from xcontour_amd import _native as nat
    with pytest.raises(nat.XContourHipError) as e:
        nat.Context(0)
    assert e.value.code in (nat.XC_ENODEV, nat.XC_EHIP)
    assert 'no CPU fallback' in str(e.value) or 'HIP' in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'xcontour_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert 'xcontour_oracle' not in txt and 'import oracle' not in txt, f
