"""CPU: the C-ABI library builds, loads and exports every symbol include/fwdtaco_hip.h declares."""
import ctypes
import os

import pytest

from forwardtacotron_amd import _lib, build


@pytest.fixture(scope='module')
def built():
    return build.build(verbose=False)


def test_header_parses():
    protos = _lib.parse_header()
    assert 'ft_last_error' in protos and 'ft_lr_expand' in protos and 'ft_conv_bank_fwd' in protos
    for name, (ret, args) in protos.items():
        assert ret in ('int', 'size_t', 'const char*', 'void'), (name, ret)
        for t, n in args:
            _lib._ctype(t)        # every argument type is bindable


def test_library_exports_all_declared_symbols(built):
    L = ctypes.CDLL(built)
    for name in _lib.parse_header():
        assert hasattr(L, name), f'{name} declared in fwdtaco_hip.h but not exported'


def test_binding_loads_and_reports_abi(built):
    L = _lib.lib()
    assert L.ft_abi_version() >= 1
    assert L.ft_last_error() is not None


def test_no_torch_types_in_abi():
    protos = _lib.parse_header()
    for name, (ret, args) in protos.items():
        for t, _ in args:
            assert 'Tensor' not in t and 'at::' not in t and 'torch' not in t, (name, t)
