"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/streammind_hip.h
declares (no compute without a GPU), and the ctypes structs match the C structs."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "streammind_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from streammind_amd import _lib
    lib = _lib.load()                       # raises if the .so is missing: no fallback on the product path
    decl = _declared_symbols()
    assert len(decl) >= 35
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/streammind_hip.h but not exported"
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    assert lib.sm_abi_version() == 4
    assert lib.sm_packed_elems(40, 70) == 3 * 3 * 512


def test_struct_layouts_match_header(tmp_path):
    """compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    import ctypes as C
    import subprocess
    from streammind_amd._lib import sm_linear_t, sm_config_t
    src = tmp_path / "sz.c"
    fields_l = [f[0] for f in sm_linear_t._fields_]
    fields_c = [f[0] for f in sm_config_t._fields_]
    body = "".join(f'printf("%zu\\n", offsetof(sm_linear_t, {f}));' for f in fields_l)
    body += "".join(f'printf("%zu\\n", offsetof(sm_config_t, {f}));' for f in fields_c)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "streammind_hip.h"\nint main(){'
                   'printf("%zu\\n%zu\\n", sizeof(sm_linear_t), sizeof(sm_config_t));' + body + 'return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert vals[0] == C.sizeof(sm_linear_t) and vals[1] == C.sizeof(sm_config_t)
    want = [getattr(sm_linear_t, f).offset for f in fields_l] + [getattr(sm_config_t, f).offset for f in fields_c]
    assert vals[2:] == want


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    from streammind_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
        raise AssertionError("expected failure")
    except _lib.StreamMindHipError as e:
        assert "no CPU fallback" in str(e)
    importlib.reload(_lib)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "streammind_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} references the oracle"


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        return
    from streammind_amd._lib import StreamMindHipError
    from streammind_amd.native import NativeModel, PathConfig
    try:
        NativeModel(PathConfig(llm_layers=0))
        raise AssertionError("expected failure without a GPU")
    except StreamMindHipError as e:
        assert "no HIP device" in str(e)


def test_torch_library_ops_are_registered_with_fake_kernels():
    """the C ABI re-exposed as torch.ops.streammind_hip.* (PyTorch-ROCm custom ops); shape inference through the fake
    kernels works without a GPU (the real kernels need one)."""
    import torch
    import streammind_amd.torch_ops as T
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in T.OPS:
        assert hasattr(torch.ops.streammind_hip, name), name
    with FakeTensorMode():
        x = torch.empty(577, 1024, dtype=torch.bfloat16)
        y = torch.ops.streammind_hip.linear(x, torch.empty(8, dtype=torch.bfloat16), 3072, 1024, None, 1, None, torch.bfloat16)
        assert tuple(y.shape) == (577, 3072) and y.dtype == torch.bfloat16
        st = T.new_stream_state("cpu")
        lg, dc = torch.ops.streammind_hip.stream_push_frames(0, torch.empty(4, 336, 336, 3, dtype=torch.uint8), st)
        assert tuple(lg.shape) == (4, 2) and dc.dtype == torch.int32
        assert tuple(torch.ops.streammind_hip.llm_decode(0, 5, st).shape) == (5,)
        assert tuple(torch.ops.streammind_hip.vit_attention(torch.empty(2 * 577, 3072, dtype=torch.float16), 2, 577, 16, 64).shape) == (1154, 1024)
        assert tuple(torch.ops.streammind_hip.pool_rows(torch.empty(5, 576, 1024)).shape) == (5, 1024)
    # the ops that advance a stream behind its handle declare the state tensor as mutated: graph capture can neither drop, merge
    # nor reorder them (the operator-level ops stay pure)
    for name in ("stream_push_frames", "stream_push_pooled", "llm_prefill", "llm_decode"):
        schema = getattr(torch.ops.streammind_hip, name).default._schema
        assert any(a.name == "state" and a.alias_info is not None and a.alias_info.is_write for a in schema.arguments), name
    assert not any(a.alias_info is not None and a.alias_info.is_write for a in torch.ops.streammind_hip.linear.default._schema.arguments)
