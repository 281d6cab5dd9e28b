"""TorchScript export surface of ForwardTacotron (reference: models/forward_tacotron.py:186-200 `generate_jit`,
README.md:159-171 "Export Model with TorchScript").

The reference's snippet is

    tts_model.eval(); model_script = torch.jit.script(tts_model); y = model_script.generate_jit(x)

and it works unchanged on the drop-in module.  What gets scripted is not the nn.Module tree (its forward paths are
sequences of hand-written gfx950 kernels reached through a C ABI -- nothing TorchScript could compile) but the module
`ScriptedForwardTacotron` that `ForwardTacotron.__prepare_scriptable__` returns:

  * every parameter and float buffer of the model in ONE flat fp32 buffer (`flat_weights`, a registered buffer: it
    follows `.cuda()` / `.to()` and is what `torch.jit.save` serialises), plus a JSON layout (name, shape, offset) and
    the constructor kwargs as JSON strings;
  * `generate_jit(x, alpha, beta)` (and `forward(x)`) = one call of the opaque operator
    `torch.ops.fwdtaco.generate_jit`, registered below with `torch.library`.  Its implementation rebuilds a
    ForwardTacotron whose parameters are zero-copy VIEWS into the flat buffer (cached per buffer address) and runs the
    same HIP mel-generation path as the eager `generate_jit`.

`torch.jit.save(model_script, path)` / `torch.jit.load(path)` round-trip; the loading process must have imported
`forwardtacotron_amd.export` (or the package's model module, which imports it) so that the operator exists -- the
artifact is Python-free in the TorchScript sense (no pickled classes), not library-free: it needs libfwdtaco_hip.so and
an MI355X exactly like the eager module.  The operator has no CPU kernel and raises on CPU tensors.
"""
import json
from typing import Dict, List

import torch
import torch.nn as nn

_LIB = torch.library.Library('fwdtaco', 'DEF')
_LIB.define('generate_jit(Tensor x, float alpha, float beta, Tensor flat_weights, str cfg, str layout) -> Tensor[]')

_OUT_KEYS = ('mel', 'mel_post', 'dur', 'pitch', 'energy')
_models = {}


def _rebuild(flat: torch.Tensor, cfg: str, layout: str):
    """ForwardTacotron whose float tensors alias `flat`.  The cache holds at most TWO rebuilt models (their parameters
    pin the flat buffer they alias: a scripted module that was deleted or moved must not stay on the GPU behind a
    forgotten entry) and compares hashes, not the JSON strings."""
    key = (flat.data_ptr(), flat.device, hash(cfg), hash(layout))
    m = _models.get(key)
    if m is None:
        from .model import ForwardTacotron
        while len(_models) >= 2:
            _models.pop(next(iter(_models)))          # oldest first (dicts keep insertion order)
        with torch.device('meta'):            # no allocation, no initialisation: every float tensor is re-pointed below
            m = ForwardTacotron(**json.loads(cfg))
        def owner(name):
            mod = m
            *path, leaf = name.split('.')
            for p in path:
                mod = getattr(mod, p)
            return mod, leaf

        for name, shape, off in json.loads(layout):
            n = 1
            for d in shape:
                n *= d
            view = flat[off:off + n].view(shape)
            mod, leaf = owner(name)
            if leaf in mod._parameters:
                mod._parameters[leaf] = nn.Parameter(view, requires_grad=False)
            else:
                mod._buffers[leaf] = view
        for name, b in list(m.named_buffers()):     # integer buffers (step, num_batches_tracked): unused by generation
            if b.is_meta:
                mod, leaf = owner(name)
                mod._buffers[leaf] = torch.zeros(b.shape, dtype=b.dtype, device=flat.device)
        left = [n for n, t in list(m.named_parameters()) + list(m.named_buffers()) if t.is_meta]
        if left:
            raise RuntimeError(f'scripted ForwardTacotron: the layout does not cover {left[:3]} ...')
        m.eval()
        _models[key] = m
    return m


def _generate_jit_hip(x: torch.Tensor, alpha: float, beta: float, flat_weights: torch.Tensor, cfg: str,
                      layout: str) -> List[torch.Tensor]:
    m = _rebuild(flat_weights, cfg, layout)
    with torch.no_grad():
        out = m.generate_jit(x, alpha, beta)
    return [out[k] for k in _OUT_KEYS]


def _generate_jit_cpu(x, alpha, beta, flat_weights, cfg, layout):
    raise RuntimeError('fwdtaco::generate_jit runs on an MI355X (HIP) device only: move the scripted module and the '
                       'input with .cuda(); there is no CPU fallback')


_LIB.impl('generate_jit', _generate_jit_hip, 'CUDA')
_LIB.impl('generate_jit', _generate_jit_cpu, 'CPU')


class ScriptedForwardTacotron(nn.Module):
    """What torch.jit.script(ForwardTacotron) compiles (see the module docstring)."""

    def __init__(self, flat_weights: torch.Tensor, cfg: str, layout: str) -> None:
        super().__init__()
        self.register_buffer('flat_weights', flat_weights)
        self.cfg = cfg
        self.layout = layout

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self.generate_jit(x, 1.0, 1.0)

    @torch.jit.export
    def generate_jit(self, x: torch.Tensor, alpha: float = 1.0, beta: float = 1.0) -> Dict[str, torch.Tensor]:
        with torch.no_grad():
            o = torch.ops.fwdtaco.generate_jit(x, alpha, beta, self.flat_weights, self.cfg, self.layout)
            return {'mel': o[0], 'mel_post': o[1], 'dur': o[2], 'pitch': o[3], 'energy': o[4]}


def scriptable(model: nn.Module, ctor_kwargs: dict) -> ScriptedForwardTacotron:
    """Flat-buffer twin of `model` (a copy of its current float tensors; later updates of `model` do not reach it)."""
    entries, chunks, off = [], [], 0
    for name, t in list(model.named_parameters()) + list(model.named_buffers()):
        if not t.dtype.is_floating_point:
            continue
        entries.append([name, list(t.shape), off])
        chunks.append(t.detach().reshape(-1).to(torch.float32))
        off += (t.numel() + 3) // 4 * 4                 # 16-byte aligned starts, like parallel.FlatParams
        pad = off - sum(c.numel() for c in chunks)
        if pad:
            chunks.append(torch.zeros(pad, dtype=torch.float32, device=t.device))
    flat = torch.cat(chunks) if chunks else torch.zeros(0)
    return ScriptedForwardTacotron(flat, json.dumps(ctor_kwargs), json.dumps(entries))
