"""Import plumbing that lets the reference (xinding-sys/StreamMind, read-only at
/root/reference) be imported in THIS container so it can pin the oracle.

TEST INFRASTRUCTURE ONLY.  This file contains no reference source -- only
`sys.modules` stubs for third-party packages that are not installed here and
that the reference imports at module scope but never executes on the streaming
path (decord, timm, lightning, ...), plus the one piece of un-vendored
third-party code the path does execute: `mamba_ssm.models.mixer_seq_simple.
create_block` (mamba-ssm 2.2.2, requirements.txt:156), restated below from its
published defaults (Mamba1 mixer, nn.LayerNorm eps=1e-5, fused_add_norm=False,
residual_in_fp32=False, no MLP when d_intermediate == 0).

The reference never travels to the GPU box; `available()` is False there and
every consumer must skip.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types
from functools import partial

REF_ROOT = "/root/reference"
REF_PKG = os.path.join(REF_ROOT, "streammind")

_installed = False


def available() -> bool:
    return os.path.isdir(REF_PKG)


def _stub(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []  # behave as a package so sub-imports resolve through sys.modules
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install() -> None:
    """Make `import streammind...` / `import videollama2...` resolve to the reference."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present (expected on the GPU box); skip")
    import torch
    import torch.nn as nn

    # ---- 1. the reference package, without running its eager __init__
    pkg = types.ModuleType("streammind")
    pkg.__path__ = [REF_PKG]
    pkg.__spec__ = importlib.machinery.ModuleSpec("streammind", loader=None, is_package=True)
    sys.modules["streammind"] = pkg
    sys.modules["videollama2"] = pkg  # every absolute import in the tree says videollama2.*

    class _Alias(importlib.abc.MetaPathFinder):
        """videollama2.x.y -> streammind.x.y (same module object)."""

        def find_spec(self, fullname, path=None, target=None):
            if not fullname.startswith("videollama2."):
                return None
            real = "streammind." + fullname[len("videollama2."):]
            mod = importlib.import_module(real)
            sys.modules[fullname] = mod
            return importlib.machinery.ModuleSpec(fullname, _AliasLoader(mod))

    class _AliasLoader(importlib.abc.Loader):
        def __init__(self, mod):
            self.mod = mod

        def create_module(self, spec):
            return self.mod

        def exec_module(self, module):
            pass

    sys.meta_path.insert(0, _Alias())

    # ---- 2. third-party packages that are absent here and dead on the streaming path
    class _Dummy:
        def __init__(self, *a, **k):
            pass

    _stub("decord", VideoReader=_Dummy, cpu=lambda *a, **k: None)
    _stub("imageio")
    _stub("moviepy")
    _stub("moviepy.editor", VideoFileClip=_Dummy)
    _stub("cv2")
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.regnet", RegStage=_Dummy)
    _stub("timm.models.layers", LayerNorm=nn.LayerNorm, LayerNorm2d=nn.LayerNorm)

    class _Lightning(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    _stub("pytorch_lightning", LightningModule=_Lightning)
    _stub("lightning", LightningModule=_Lightning)
    _stub("lightning.pytorch")
    _stub("lightning.pytorch.callbacks", LearningRateMonitor=_Dummy)
    _stub("torchmetrics")
    _stub("torchmetrics.functional", accuracy=lambda *a, **k: None)
    _stub("selective_scan_cuda")
    _stub("causal_conv1d")  # import of its symbols raises ImportError -> reference falls back (mamba_simple.py:14-17)

    # ---- 3. the vendored (dead) mamba_ssm copy as top-level `mamba_ssm`; Triton bits stubbed
    ms = types.ModuleType("mamba_ssm")
    ms.__path__ = [os.path.join(REF_PKG, "model", "mamba_ssm")]
    ms.__spec__ = importlib.machinery.ModuleSpec("mamba_ssm", loader=None, is_package=True)
    sys.modules["mamba_ssm"] = ms
    for sub in ("ops", "modules", "utils"):
        m = types.ModuleType(f"mamba_ssm.{sub}")
        m.__path__ = [os.path.join(REF_PKG, "model", "mamba_ssm", sub)]
        m.__spec__ = importlib.machinery.ModuleSpec(f"mamba_ssm.{sub}", loader=None, is_package=True)
        sys.modules[f"mamba_ssm.{sub}"] = m
    _stub("mamba_ssm.ops.triton")
    _stub("mamba_ssm.ops.triton.layer_norm", RMSNorm=None, layer_norm_fn=None, rms_norm_fn=None)
    _stub("mamba_ssm.ops.triton.selective_state_update", selective_state_update=None)

    ssi = importlib.import_module("mamba_ssm.ops.selective_scan_interface")
    simple = importlib.import_module("mamba_ssm.modules.mamba_simple")
    # CUDA scan -> the reference's own pure-torch spec (selective_scan_interface.py:91-157)
    simple.selective_scan_fn = ssi.selective_scan_ref
    simple.causal_conv1d_fn = None
    simple.causal_conv1d_update = None
    simple.selective_state_update = None
    block = importlib.import_module("mamba_ssm.modules.block")

    # ---- 4. mamba_ssm.models.mixer_seq_simple (NOT vendored): restated from mamba-ssm 2.2.2 defaults
    def create_block(d_model, d_intermediate=0, ssm_cfg=None, attn_layer_idx=None, attn_cfg=None,
                     norm_epsilon=1e-5, rms_norm=False, residual_in_fp32=False, fused_add_norm=False,
                     layer_idx=None, device=None, dtype=None):
        assert d_intermediate == 0 and not rms_norm and not fused_add_norm
        mixer_cls = partial(simple.Mamba, layer_idx=layer_idx, **(ssm_cfg or {}))
        norm_cls = partial(nn.LayerNorm, eps=norm_epsilon)
        blk = block.Block(d_model, mixer_cls, nn.Identity, norm_cls=norm_cls,
                          fused_add_norm=False, residual_in_fp32=residual_in_fp32)
        blk.layer_idx = layer_idx
        return blk

    def _init_weights(module, n_layer, **k):  # only affects random init; weights are overwritten
        return

    _stub("mamba_ssm.models")
    _stub("mamba_ssm.models.mixer_seq_simple", create_block=create_block, _init_weights=_init_weights)

    # ---- 5. transformers-5 drift (reference pins 4.44.2, requirements.txt:355)
    import transformers
    from transformers import MistralForCausalLM, CLIPVisionModel  # noqa: F401  force lazy modules
    import transformers.generation  # noqa: F401
    tr = sys.modules["transformers"]
    if not hasattr(tr, "TRANSFORMERS_CACHE"):
        try:
            tr.TRANSFORMERS_CACHE = os.path.expanduser("~/.cache/huggingface/hub")
        except Exception:
            pass
    _installed = True


def import_ref(name: str):
    """import a reference module by its in-tree dotted name, e.g. 'model.multimodal_projector.builder'."""
    install()
    return importlib.import_module("streammind." + name)
