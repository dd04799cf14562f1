"""CPU-side tests: the C-ABI library loads and exports every declared symbol,
and the Keras-compatible host surface behaves like the reference's classes
(constructor arguments, build(), weight layouts, get_config, error behaviour).
No compute call is made here (there is no GPU and no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import qnn_amd
from qnn_amd import _abi, nets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "qnn_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qnn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _declared_symbols()
    assert "qnn_conv2d_forward" in names and "qnn_pack_f32" in names
    lib = ctypes.CDLL(_abi.lib_path())
    for n in names:
        assert hasattr(lib, n), "libqnn_hip.so does not export %s" % n
    assert sorted(names) == sorted(_abi.EXPORTS)
    assert _abi.load().qnn_version() == 4


def test_ctypes_structures_match_the_header_layout(tmp_path):
    """The structs a binding passes by pointer (qnn_epilogue_t, qnn_projection_t, qnn_fold_info_t): size and every field offset
    of the ctypes mirrors in _abi.py against what the C compiler makes of include/qnn_abi.h."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    mirrors = {"qnn_epilogue_t": _abi.Epilogue, "qnn_projection_t": _abi.Projection, "qnn_fold_info_t": _abi.FoldInfo}
    lines = []
    for cname, cls in mirrors.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src = tmp_path / "layout.c"
    src.write_text("#include <stdio.h>\n#include <stddef.h>\n#include \"qnn_abi.h\"\nint main(void) {\n%s\nreturn 0; }\n"
                   % "\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {}
    for ln in filter(None, out):
        a, b, c = ln.split()
        got[(a, b)] = int(c)
    for cname, cls in mirrors.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
    # every field of the C structs is mirrored (a field added to the header only would shift nothing but be missing here)
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "qnn_abi.h")).read(), flags=re.S)
    for cname, cls in mirrors.items():
        body = re.search(r"typedef struct \w+ \{([^{}]*)\} %s;" % cname, hdr).group(1)
        nfields = sum(len(decl.split(",")) for decl in body.split(";") if decl.strip())
        assert nfields == len(cls._fields_), (cname, nfields, len(cls._fields_))


def test_no_cpu_fallback():
    x = torch.zeros(4, 4)
    with pytest.raises(_abi.QnnError):
        qnn_amd.binary_tanh(x)
    with pytest.raises(_abi.QnnError):
        qnn_amd.quantized_tanh(x, 4)
    layer = qnn_amd.BinaryConv2D(8, kernel_size=(3, 3), padding="same", device="cpu")
    with pytest.raises(_abi.QnnError):
        layer(torch.zeros(1, 4, 4, 3))


def test_conv_ctor_surface_and_build():
    # binary_layers.py:100-158 / quantized_layers.py:104-162
    l = qnn_amd.BinaryConv2D(64, kernel_size=(3, 3), strides=(1, 1), padding="same",
                             kernel_initializer="glorot_uniform", kernel_regularizer=None,
                             input_shape=(32, 32, 3), device="cpu")
    assert l.H == 1.0 and l.kernel_lr_multiplier == "Glorot" and l.bias_lr_multiplier is None
    l.build((None, 32, 32, 3))
    assert tuple(l.kernel.shape) == (3, 3, 3, 64) and tuple(l.bias.shape) == (64,)
    assert isinstance(l.kernel_lr_multiplier, np.float32)
    assert l.kernel_lr_multiplier == np.float32(20.0499382)
    assert float(l.kernel.abs().max()) <= 1.0
    assert l.lr_multipliers == [l.kernel_lr_multiplier, None]
    assert l.compute_output_shape((None, 32, 32, 3)) == (None, 32, 32, 64)
    q = qnn_amd.QuantizedConv2D(16, kernel_size=3, strides=2, padding="same", use_bias=False,
                                nb=4, H=1, device="cpu")
    q.build((None, 32, 32, 16))
    assert q.bias is None and q.lr_multipliers == [q.kernel_lr_multiplier]
    assert q.compute_output_shape((None, 32, 32, 16)) == (None, 16, 16, 16)
    cfg = q.get_config()
    assert cfg["nb"] == 4 and cfg["H"] == 1 and cfg["strides"] == (2, 2)
    with pytest.raises(ValueError):
        qnn_amd.BinaryConv2D(8, kernel_size=3, device="cpu").build((None, 8, 8, None))


def test_dense_ctor_surface_and_weights_roundtrip():
    d = qnn_amd.QuantizedDense(10, nb=4, device="cpu")   # units positional (vgg.py:41)
    d.build((None, 1024))
    assert tuple(d.kernel.shape) == (1024, 10)
    assert d.kernel_lr_multiplier == np.float32(1.0 / np.sqrt(1.5 / (1024 + 10)))
    k = np.random.default_rng(0).uniform(-1, 1, (1024, 10)).astype(np.float32)
    b = np.arange(10, dtype=np.float32)
    d.set_weights([k, b])
    k2, b2 = d.get_weights()
    np.testing.assert_array_equal(k, k2)
    np.testing.assert_array_equal(b, b2)
    with pytest.raises(ValueError):
        d.set_weights([k])
    with pytest.raises(ValueError):
        d.set_weights([k[:10], b])
    bd = qnn_amd.BinaryDense(10, H="Glorot", device="cpu")
    bd.build((None, 64))
    assert bd.H == np.float32(np.sqrt(1.5 / 74))
    with pytest.raises(AssertionError):
        qnn_amd.BinaryDense(3, device="cpu").build((5,))


def test_aliases():
    assert qnn_amd.BinaryConvolution2D is qnn_amd.BinaryConv2D
    assert qnn_amd.QuantizedConvolution2D is qnn_amd.QuantizedConv2D
    assert qnn_amd.TernaryConvolution2D is qnn_amd.TernaryConv2D


def test_ternary_ctor_surface():
    # ternary_layers.py:37-41,100-106
    t = qnn_amd.TernaryConv2D(32, kernel_size=(3, 3), padding="same", H=1., device="cpu")
    t.build((None, 8, 8, 16))
    assert tuple(t.kernel.shape) == (3, 3, 16, 32) and t.kernel_lr_multiplier == np.float32(1. / np.sqrt(1.5 / (144 + 288)))
    d = qnn_amd.TernaryDense(10, device="cpu")
    d.build((None, 64))
    assert set(d.get_config()) >= {"H", "kernel_lr_multiplier", "bias_lr_multiplier", "units"}
    for nt in ("tnn", "qtnn", "full-tnn"):
        spec = nets.build_spec(nets.Config(network_type=nt, architecture="VGG"), 1)
        assert all(op["kind"] == "ternary" for op in spec if op["op"] in ("conv", "dense"))


def test_baseline_specs_shapes():
    macs = {}
    for idx in range(5):
        cf = nets.baseline_config(idx)
        if idx == 4:
            cf.dim = 32   # same topology, CIFAR-sized, to keep the test light
        spec = nets.build_spec(cf, nets.SEED_BASE + idx)
        convs = [op for op in spec if op["op"] == "conv"]
        dense = [op for op in spec if op["op"] == "dense"]
        assert len(dense) == 1
        macs[idx] = len(convs)
    assert macs == {0: 3, 1: 3, 2: 3, 3: 9, 4: 63}
    cf = nets.baseline_config(2)
    spec = nets.build_spec(cf, 1)
    assert spec[0]["kernel"].shape == (3, 3, 3, 64) and spec[0]["nb"] == 4
    d = [op for op in spec if op["op"] == "dense"][0]
    assert d["kernel"].shape == (1024, 10) and d["nb"] == cf.abits   # model_factory.py:31 quirk
    assert spec[1]["eps"] == 1e-4
    x = nets.synthetic_images(cf, 3, 7)
    assert x.shape == (3, 32, 32, 3) and x.dtype == np.float32 and x.max() <= 1.0
    assert np.array_equal(np.rint(x * 255), x * np.float32(255))


def test_spec_from_reference_checkpoint_runs_in_oracle():
    """tools/import_keras_hdf5.py + nets.spec_from_keras_npz on the reference's trained
    weights_44.hdf5: 21 quantized convs (use_bias=True), 19 BN, 9 adds, softmax dense."""
    from oracle import qnn_oracle as O
    path = os.path.join(ROOT, "tests", "golden", "resnet3_full_44.npz")
    spec = nets.spec_from_keras_npz(path, 4, 4)
    kinds = [op["op"] for op in spec]
    assert kinds.count("conv") == 21 and kinds.count("bn") == 19 and kinds.count("add") == 9
    assert kinds[-1] == "softmax" and "scale" not in kinds
    assert all(op["bias"] is not None for op in spec if op["op"] == "conv")
    y = O.run_spec(spec, nets.synthetic_images(nets.Config(dim=32), 2, 3))
    assert y.shape == (2, 10) and np.allclose(y.sum(-1), 1.0, atol=1e-5)


def test_fused_chain_grouping_and_residual_planning():
    """Host-side planning (no GPU): FusedModel groups conv/bn/act/pool chains; the residual
    planner finds producers, consumers and the storage every activation is wanted in."""
    from qnn_amd import engine
    cf = nets.baseline_config(2)
    groups = engine.FusedModel._group(nets.build_spec(cf, 1))
    assert [g["kind"] for g in groups] == ["conv", "conv", "conv", "dense"]
    assert [g["pool"] for g in groups] == [2, 2, 2, 1]
    assert all(g["bn"] is not None for g in groups) and groups[-1]["act"] is None
    rcf = nets.Config(network_type="full-qnn", wbits=4, abits=4, architecture="RESNET", nres=1, dim=32)
    rspec = nets.build_spec(rcf, 2)
    assert engine.FusedModel._group(rspec) is None          # residual topology is not a chain
    m = engine.ResidualFusedModel(rspec, device="cpu")
    acts = [n for n, i in m.prod.items() if rspec[i]["op"] == "act"]
    stores = {m._act_out_store(n, 4) for n in acts}
    assert stores == {_abi.STORE_I4}                        # every activation stays packed: the avg-pool reads the codes too
    adds = [i for i, op in enumerate(rspec) if op["op"] == "add"]
    assert len(adds) == 3 and all(len(m.srcs[i]) == 2 for i in adds)
    bcf = nets.Config(network_type="full-bnn", architecture="RESNET", nres=1, dim=32)
    mb = engine.ResidualFusedModel(nets.build_spec(bcf, 2), device="cpu")
    assert _abi.STORE_BIN in {mb._act_out_store(n, 1) for n, i in mb.prod.items() if mb.spec[i]["op"] == "act"}


def test_one_bit_layers_choose_the_matrix_pipe_by_shape_and_impl():
    """engine._matrix_pipe_1bit: 3x3, Cin in {64,128}, Cout == 64 binary convs go to the int8
    MFMA kernel (int4-stored +-1 codes) unless the VALU-only family is selected."""
    from qnn_amd import _abi, engine
    k = lambda kh, cin, cout: np.zeros((kh, kh, cin, cout), np.float32)
    conv = lambda kind, kern: {"op": "conv", "kind": kind, "kernel": kern}
    saved = _abi._conv_impl
    try:
        _abi._conv_impl = _abi.IMPL_AUTO
        assert engine._matrix_pipe_1bit(conv("binary", k(3, 64, 64)))
        assert engine._matrix_pipe_1bit(conv("binary", k(3, 128, 64)))
        assert not engine._matrix_pipe_1bit(conv("binary", k(3, 64, 128)))      # more than one filter slice
        assert not engine._matrix_pipe_1bit(conv("binary", k(1, 64, 64)))       # 1x1
        assert not engine._matrix_pipe_1bit(conv("binary", k(3, 32, 64)))       # Cin not a multiple of 64
        assert not engine._matrix_pipe_1bit(conv("quantized", k(3, 64, 64)))    # not a 1-bit layer
        assert not engine._matrix_pipe_1bit({"op": "dense", "kind": "binary", "kernel": np.zeros((64, 10), np.float32)})
        assert not engine._matrix_pipe_1bit(None)
        _abi._conv_impl = _abi.IMPL_VALU
        assert not engine._matrix_pipe_1bit(conv("binary", k(3, 64, 64)))
        assert _abi.conv_impl() == _abi.IMPL_VALU
    finally:
        _abi._conv_impl = saved


def test_default_library_carries_no_measurement_scaffolding():
    """The A/B environment switches and the GEMM's operand ablations are compiled in only with -DQNN_EXPERIMENTS
    (tools/build_variant.py): the shipped library must not even contain their names."""
    import importlib
    b = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd._build")
    blob = open(b.LIB, "rb").read()
    for name in (b"QNN_MFMA_ABLATE", b"QNN_FIRST_ABL", b"QNN_MFMA_TILE", b"QNN_MFMA_SHAPE", b"QNN_MFMA_DMA",
                 b"QNN_MFMA_AREG", b"QNN_MFMA_WRES", b"QNN_MFMA_SMALL_OFF", b"QNN_FIRST_GATHER", b"QNN_FIXED_BPC",
                 b"QNN_XNOR_TR"):
        assert name not in blob, name
    assert "-DQNN_EXPERIMENTS" not in " ".join(b.CFLAGS)
