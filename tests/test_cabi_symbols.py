"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mfg_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'mfg_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mfg_[a-zA-Z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    from discrete_mean_field_game_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def test_header_and_binding_agree(lib):
    decl = declared_symbols()
    assert len(decl) >= 18
    assert sorted(lib.SIGNATURES) == decl


def test_every_symbol_is_exported(lib):
    handle = lib.lib()
    for name in declared_symbols():
        assert getattr(handle, name) is not None


def test_host_only_helpers(lib):
    """Integer bookkeeping exported by the library is bit exact with the oracle (no GPU needed)."""
    from oracle import mfg_oracle as O
    h = lib.lib()
    assert h.mfg_abi_version() == 17
    for d in (1, 3, 4, 21, 47, 128, 256):
        assert h.mfg_num_features(d) == O.num_features(d)
        for i in range(0, d, max(1, d // 7)):
            for j in range(0, d, max(1, d // 5)):
                assert h.mfg_feature_index(i, j, d) == O.feature_index(i, j, d)
    assert h.mfg_workspace_bytes(65536 * 15, 21) >= (O.num_features(21) + 3) * 8
    assert h.mfg_workspace_bytes(0, 21) == 0


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'LIB_PATH', '/nonexistent/libmfg_hip.so')
    with pytest.raises(lib.MfgError):
        lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'discrete_mean_field_game_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'mfg_oracle' not in src, f


def test_reward_net_struct_layout_matches_the_header(lib, tmp_path):
    """mfg_reward_net_t crosses the boundary by pointer: the ctypes mirror (_lib.RewardNetStruct) must have the C
    compiler's size and field offsets.  A three-line C program prints them from include/mfg_hip.h."""
    import ctypes as C
    import subprocess
    fields = [n for n, _ in lib.RewardNetStruct._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "mfg_hip.h"\nint main(void){printf("%zu", sizeof(mfg_reward_net_t));\n'
    for f in fields:
        src += 'printf(" %%zu", offsetof(mfg_reward_net_t, %s));\n' % f
    src += 'return 0;}\n'
    c = tmp_path / 'layout.c'
    c.write_text(src)
    exe = str(tmp_path / 'layout')
    subprocess.run(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'), str(c), '-o', exe], check=True)
    out = [int(x) for x in subprocess.run([exe], check=True, stdout=subprocess.PIPE).stdout.decode().split()]
    assert out[0] == C.sizeof(lib.RewardNetStruct)
    assert out[1:] == [getattr(lib.RewardNetStruct, f).offset for f in fields]
