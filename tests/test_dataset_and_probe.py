"""The accuracy-harness pieces of SURVEY 8(f4): the dataset loader of utils/load_data.py (from a local .npz instead of
a download) and the |activation| > 63 probe of test_resnet.py:70-89."""
import numpy as np
import pytest

import qnn_amd  # noqa: F401
from qnn_amd import nets

F32 = np.float32


def _write_npz(tmp_path, n_train, n_test, shape, seed=0):
    rng = np.random.default_rng(seed)
    p = tmp_path / "data.npz"
    np.savez(p, x_train=rng.integers(0, 256, (n_train,) + shape, dtype=np.uint8),
             y_train=rng.integers(0, 10, (n_train, 1), dtype=np.uint8),
             x_test=rng.integers(0, 256, (n_test,) + shape, dtype=np.uint8),
             y_test=rng.integers(0, 10, (n_test, 1), dtype=np.uint8))
    return str(p)


def test_load_dataset_cifar_layout_and_hinge_targets(tmp_path):
    p = _write_npz(tmp_path, 45010, 7, (4, 4, 3))            # tiny images, the real split point
    train, valid, test = nets.load_dataset("CIFAR-10", p, architecture="VGG")
    assert train.X.shape == (45000, 4, 4, 3) and valid.X.shape == (10, 4, 4, 3) and test.X.shape == (7, 4, 4, 3)
    assert train.X.dtype == np.float32 and 0.0 <= train.X.min() and train.X.max() <= 1.0
    raw = np.load(p)
    np.testing.assert_array_equal(test.X, raw["x_test"].astype("float32") / 255)       # load_data.py:40
    assert set(np.unique(test.y)) == {-1.0, 1.0} and test.y.shape == (7, 10)             # 2 * one-hot - 1
    np.testing.assert_array_equal(test.y.argmax(1), raw["y_test"].reshape(-1))
    _, _, test_r = nets.load_dataset("CIFAR-10", p, architecture="RESNET")
    assert set(np.unique(test_r.y)) == {0.0, 1.0}                                        # plain one-hot


def test_load_dataset_mnist_adds_the_channel_axis(tmp_path):
    p = _write_npz(tmp_path, 50004, 3, (5, 5))
    train, valid, test = nets.load_dataset("MNIST", p)
    assert train.X.shape == (50000, 5, 5, 1) and valid.X.shape == (4, 5, 5, 1) and test.X.shape == (3, 5, 5, 1)
    with pytest.raises(ValueError, match="not supported"):
        nets.load_dataset("SVHN", p)


@pytest.mark.gpu
def test_activation_range_probe_matches_the_oracle():
    from oracle import qnn_oracle as O
    cf = nets.Config(network_type="full-qnn", architecture="RESNET", dataset="CIFAR-10", dim=32, channels=3,
                     wbits=4, abits=4, nres=1, pfilt=1)
    model = nets.build_model(cf, seed=5, first_layer="exact")
    x = nets.synthetic_images(cf, 6, seed=9)
    convs = [i for i, op in enumerate(model.spec) if op["op"] == "conv"]
    number = 3
    got = model.conv_output(x, number)
    names = [op.get("dst", "t%d" % i) for i, op in enumerate(model.spec)]
    env = O.run_spec(model.spec[:convs[number - 1] + 1], x, return_all=True, float_conv="device")
    want = env[names[convs[number - 1]]] if isinstance(env, dict) else env
    tol = 1e-5 * np.maximum(1.0, np.abs(want))
    assert got.shape == want.shape and np.all(np.abs(got - want) <= tol)
    limit = float(np.quantile(np.abs(want), 0.999))           # a limit some values exceed
    rep = nets.activation_range_probe(model, x, number, limit=limit)
    over = np.abs(got) > limit
    assert rep["count"] == int(over.sum()) > 0
    assert rep["max_abs"] == pytest.approx(float(np.abs(got).max()))
    assert rep["first"][0] == int(np.argwhere(over)[0][0])
    assert nets.activation_range_probe(model, x, number, limit=1e9) == {"count": 0, "max_abs": rep["max_abs"], "first": None}
