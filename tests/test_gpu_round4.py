"""Round-4 regression tests on the GPU: the "auto" first layer (the product default: byte kernel with a per-batch domain
flag, a batch that is not image bytes / 255 recomputed on the exact kernel -- no QnnError, no silent change), the
pipelined rings bench.py's N > 1 path replays, and Model.predict's chunked upload."""
import numpy as np
import pytest
import torch

from qnn_amd import _abi, engine, nets
from oracle import qnn_oracle as O
from test_gpu_parity import dev, host

pytestmark = pytest.mark.gpu
F32 = np.float32


def _cfg(idx):
    cf = nets.baseline_config(idx)
    return cf, nets.build_spec(cf, nets.SEED_BASE + idx)


@pytest.mark.parametrize("idx", [1, 2])
def test_auto_first_layer_is_the_default_and_accepts_every_float_tensor(idx):
    """FusedModel(spec) with no arguments: dataset images (bytes / 255) give the byte kernel's bits (= the uint8 entry's
    specification), ANY other float tensor gives the exact float32 path bit for bit (the reference's call() takes every
    float, quantized_layers.py:164-194) -- and the model stays usable afterwards."""
    cf, spec = _cfg(idx)
    m = engine.FusedModel(spec)
    assert m.first_layer == "auto"
    xu8 = nets.synthetic_images_u8(cf, 8, 3)
    x = (xu8.astype(F32) / F32(255)).astype(F32)
    m.kernel_log = []
    got = host(m(dev(x)))
    assert m.kernel_log[0] == "mfma_i8_first_img255", m.kernel_log
    np.testing.assert_array_equal(got, O.run_spec_u8(spec, xu8))
    rng = np.random.default_rng(idx)
    for xf in (rng.standard_normal(x.shape).astype(F32), (x * F32(0.999)).astype(F32), np.where(x > 0.5, x, F32(1.5)).astype(F32)):
        m.kernel_log = []
        gotf = host(m(dev(xf)))
        assert m.kernel_log[0] == "mfma_i8_first_img255" and any(k.startswith("mfma_f32_first") for k in m.kernel_log), m.kernel_log
        np.testing.assert_array_equal(gotf, O.run_spec(spec, xf, float_conv="device"))
        np.testing.assert_array_equal(gotf, host(engine.FusedModel(spec, first_layer="exact")(dev(xf))))
    np.testing.assert_array_equal(host(m(dev(x))), got)          # the flag was cleared: images take the fast kernel again
    m.check_domain()


def test_model_predict_defaults_recompute_exactly_the_flagged_batches():
    """nets.Model(cf, spec).predict (no arguments) over several batches of which SOME are not image bytes / 255: every
    batch equals its own path (byte kernel / exact kernel) bit for bit, resident tensors and host arrays alike."""
    cf, spec = _cfg(2)
    B = 64
    xs, want = [], []
    rng = np.random.default_rng(9)
    for b in range(5):
        u8 = nets.synthetic_images_u8(cf, B, 40 + b)
        if b in (1, 3):
            xf = rng.uniform(0, 1, u8.shape).astype(F32)
            xs.append(xf); want.append(O.run_spec(spec, xf, float_conv="device"))
        else:
            xs.append((u8.astype(F32) / F32(255)).astype(F32)); want.append(O.run_spec_u8(spec, u8))
    tail = (nets.synthetic_images_u8(cf, 7, 77).astype(F32) / F32(255)).astype(F32)      # ragged tail, eager
    xs.append(tail); want.append(O.run_spec_u8(spec, nets.synthetic_images_u8(cf, 7, 77)))
    x, want = np.concatenate(xs), np.concatenate(want)
    m = nets.Model(cf, spec)
    assert m.engine.first_layer == "auto"
    np.testing.assert_array_equal(host(m.predict(dev(x), batch_size=B)), want)           # resident
    m.upload_batches = 2
    np.testing.assert_array_equal(m.predict(x, batch_size=B), want)                      # host array, chunked upload
    np.testing.assert_array_equal(m.predict(x, batch_size=B), want)                      # and again (lanes reused)


def test_residual_engine_auto_first_layer():
    cf = nets.Config(network_type="full-qnn", wbits=4, abits=4, architecture="RESNET", nres=1, dim=32)
    spec = nets.build_spec(cf, 3)[:-1]
    m = engine.ResidualFusedModel(spec)
    assert m.first_layer == "auto"
    xu8 = nets.synthetic_images_u8(cf, 3, 5)
    x = (xu8.astype(F32) / F32(255)).astype(F32)
    np.testing.assert_array_equal(host(m(dev(x))), O.run_spec_u8(spec, xu8))
    xf = np.random.default_rng(1).standard_normal(x.shape).astype(F32)
    np.testing.assert_array_equal(host(m(dev(xf))), O.run_spec(spec, xf, float_conv="device"))
    # through the pipeline (hipGraph lanes with a flag word each): mixed batches
    pipe = engine.Pipelined(m, lanes=2, batch_size=3)
    xx = np.concatenate([x, xf, x])
    want = np.concatenate([O.run_spec_u8(spec, xu8), O.run_spec(spec, xf, float_conv="device"), O.run_spec_u8(spec, xu8)])
    np.testing.assert_array_equal(host(pipe(dev(xx))), want)
    pipe.check_domain()


@pytest.mark.parametrize("kind", ["fused", "residual"])
def test_pipelined_ring_slots_hold_the_eager_logits(kind):
    """engine.Pipelined.lanes_for(x, slots=4, inputs=2): the rings bench.py hands to its collectives.  FusedModel: one
    hipGraph per slot whose last kernel writes straight into the slot; other engines: one graph and a copy.  Every ring
    block must equal the eager logits of the input buffer its graph reads."""
    if kind == "fused":
        cf, spec = _cfg(2)
        m, n = engine.FusedModel(spec), 32
    else:
        cf = nets.Config(network_type="full-qnn", wbits=4, abits=4, architecture="RESNET", nres=1, dim=32)
        spec = nets.build_spec(cf, 3)[:-1]
        m, n = engine.ResidualFusedModel(spec), 4
    xa = dev(nets.synthetic_images(cf, n, 1))
    xb = dev(nets.synthetic_images(cf, n, 2))
    pipe = engine.Pipelined(m, lanes=2, batch_size=n)
    lanes = pipe.lanes_for(xa, slots=4, inputs=2)
    ea, eb = host(m(xa)), host(m(xb))
    for ln in lanes:
        assert ln["direct"] == (kind == "fused")
        if len(ln["xs"]) > 1:
            ln["xs"][1].copy_(xb)
        assert bool((ln["ring"] == 0).all())             # zero-filled, never uninitialised (round 3's NaN abort)
        B = ln["y"].shape[0]
        with torch.cuda.stream(ln["stream"]):
            for j in range(4):
                if ln["direct"]:
                    ln["graphs"][j].replay()
                else:
                    ln["graph"].replay()
                    ln["ring"][j * B:(j + 1) * B].copy_(ln["y"], non_blocking=True)
        torch.cuda.synchronize()
        ring = host(ln["ring"])
        assert np.isfinite(ring).all()
        for j in range(4):
            want = eb if (ln["direct"] and j % 2 == 1) else ea
            np.testing.assert_array_equal(ring[j * B:(j + 1) * B], want, err_msg="slot %d" % j)
    pipe.check_domain()


def test_uint8_images_with_a_faithful_trick_are_refused_not_silently_changed():
    cf, spec = _cfg(2)
    m = engine.FusedModel(spec, trick="nep50")
    with pytest.raises(_abi.QnnError, match="cannot be combined with uint8"):
        m(dev(nets.synthetic_images_u8(cf, 2, 1)))
    m(dev(nets.synthetic_images(cf, 2, 1)))              # float32 images: the trick runs
