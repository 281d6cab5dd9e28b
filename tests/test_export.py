"""TorchScript export surface (reference: models/forward_tacotron.py:186-200, README.md:159-171).  CPU part: the
reference's snippet `torch.jit.script(model)` compiles, serialises and reloads; the flat buffer reproduces the
model's float tensors bit for bit; the operator refuses CPU tensors.  The GPU part (scripted generate_jit ==
eager generate_jit == golden) is tests/test_gpu_model.py::test_torchscript_generate_jit."""
import io

import pytest
import torch

from helpers import TINY


def _model():
    from forwardtacotron_amd.model import ForwardTacotron
    torch.manual_seed(5)
    return ForwardTacotron(**TINY).eval()


def test_reference_export_snippet_compiles_and_round_trips():
    from forwardtacotron_amd import export
    m = _model()
    s = torch.jit.script(m)                                   # README.md:165 of the reference, unchanged
    assert isinstance(s, torch.jit.ScriptModule) and hasattr(s, 'generate_jit')
    assert 'fwdtaco::generate_jit' in str(s.generate_jit.graph)
    buf = io.BytesIO()
    torch.jit.save(s, buf)
    buf.seek(0)
    l = torch.jit.load(buf)
    assert torch.equal(l.flat_weights, s.flat_weights) and l.cfg == s.cfg and l.layout == s.layout
    twin = export._rebuild(l.flat_weights, l.cfg, l.layout)
    sd, sd2 = m.state_dict(), twin.state_dict()
    assert list(sd.keys()) == list(sd2.keys())
    for k in sd:
        if sd[k].dtype.is_floating_point:
            assert torch.equal(sd[k], sd2[k]), k
            assert sd2[k].data_ptr() >= l.flat_weights.data_ptr()          # a view into the flat buffer, not a copy
            assert sd2[k].data_ptr() < l.flat_weights.data_ptr() + 4 * l.flat_weights.numel()
    assert not twin.training


def test_scripted_module_refuses_cpu_tensors():
    s = torch.jit.script(_model())
    with pytest.raises(Exception, match='MI355X'):
        s.generate_jit(torch.ones(1, 5).long())
