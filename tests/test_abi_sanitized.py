"""Host half of the C-ABI library under AddressSanitizer + UndefinedBehaviorSanitizer.

tools/build_sanitized.py compiles every translation unit with -fsanitize=address,undefined on the host side only
(the device code is untouched; GPU ASan does not exist on this pool).  A child process -- the ASan runtime has to be
the first library of the process -- then drives, without a GPU:
  * every argument check of the ABI (null pointers, bad shapes / kinds / widths): status QNN_EINVAL or
    QNN_EUNSUPPORTED and a message in qnn_last_error();
  * the no-device error paths: a well-formed prepack fails inside the HIP runtime, must release what it had allocated
    (the partially built handle) and report QNN_EHIP / QNN_ENOMEM -- ASan sees a double free or a leak of host memory.
A sanitizer report makes the child exit non-zero (halt_on_error), which fails the test.
"""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent(r'''
    import ctypes, os, sys
    lib = ctypes.CDLL(os.environ["QNN_LIB"])
    vp, ci, sz, fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float
    lib.qnn_last_error.restype = ctypes.c_char_p
    lib.qnn_last_kernel.restype = ctypes.c_char_p
    lib.qnn_packed_bytes.restype = sz
    lib.qnn_packed_bytes.argtypes = [ci, sz, ci]
    lib.qnn_conv2d_workspace_bytes.restype = sz
    lib.qnn_conv2d_workspace_bytes.argtypes = [vp, ci, ci, ci]
    lib.qnn_binary_tanh_f32.argtypes = [vp, vp, sz, vp]
    lib.qnn_quantized_tanh_f32.argtypes = [vp, vp, sz, ci, vp]
    lib.qnn_ternary_tanh_f32.argtypes = [vp, vp, sz, vp, vp]
    lib.qnn_ternary_abs_sum_f32.argtypes = [vp, sz, vp, vp]
    lib.qnn_ternary_apply_f32.argtypes = [vp, vp, sz, vp, vp]
    lib.qnn_pack_f32.argtypes = [vp, vp, sz, ci, ci, ci, ci, vp]
    lib.qnn_unpack_f32.argtypes = [vp, vp, sz, ci, ci, ci, vp]
    lib.qnn_prepack_weights.argtypes = [ci, ci, fl, vp, ci, ci, ci, ci, vp, ci, ci, ci, vp, ctypes.POINTER(vp)]
    lib.qnn_free_weights.argtypes = [vp]
    lib.qnn_weights_dequant.argtypes = [vp, vp, vp]
    lib.qnn_conv2d_forward.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]
    lib.qnn_dense_forward.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp]
    lib.qnn_conv2d_forward_f32in.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, sz, vp]
    EINVAL, EUNSUP, EHIP, ENOMEM = -1, -2, -3, -4
    FAKE = 0x10000            # a non-null "device pointer": argument checks never dereference it
    n = [0]

    def expect(rc, want, what):
        n[0] += 1
        msg = lib.qnn_last_error().decode()
        ok = rc in want if isinstance(want, tuple) else rc == want
        assert ok, "%s: rc=%d, expected %s (%s)" % (what, rc, want, msg)
        if rc != 0:
            assert msg, what + ": no message in qnn_last_error()"
        return msg

    assert lib.qnn_version() == 4
    expect(lib.qnn_set_conv_impl(5), EINVAL, "set_conv_impl(5)")
    expect(lib.qnn_set_conv_impl(-1), EINVAL, "set_conv_impl(-1)")
    expect(lib.qnn_set_conv_impl(0), 0, "set_conv_impl(0)")
    assert not hasattr(lib, "qnn_set_option")          # round 4: no process-wide kernel switches in the library
    # elementwise clips
    expect(lib.qnn_binary_tanh_f32(None, FAKE, 10, None), EINVAL, "binary_tanh null x")
    expect(lib.qnn_binary_tanh_f32(FAKE, None, 10, None), EINVAL, "binary_tanh null y")
    expect(lib.qnn_binary_tanh_f32(FAKE, FAKE, 0, None), 0, "binary_tanh n=0")
    expect(lib.qnn_quantized_tanh_f32(FAKE, FAKE, 10, 0, None), EINVAL, "quantized_tanh nb=0")
    expect(lib.qnn_quantized_tanh_f32(FAKE, FAKE, 10, 25, None), EINVAL, "quantized_tanh nb=25")
    expect(lib.qnn_quantized_tanh_f32(None, None, 10, 4, None), EINVAL, "quantized_tanh null")
    expect(lib.qnn_ternary_tanh_f32(FAKE, FAKE, 10, None, None), EINVAL, "ternary_tanh null workspace")
    expect(lib.qnn_ternary_abs_sum_f32(None, 10, FAKE, None), EINVAL, "ternary_abs_sum null")
    expect(lib.qnn_ternary_apply_f32(FAKE, None, 10, FAKE, None), EINVAL, "ternary_apply null")
    expect(lib.qnn_ternary_tanh_f32(FAKE, FAKE, 0, FAKE, None), 0, "ternary_tanh n=0")
    # packing
    assert lib.qnn_packed_bytes(1, 10, 64) == 10 * 2 * 4 and lib.qnn_packed_bytes(4, 3, 20) == 3 * 3 * 4
    assert lib.qnn_packed_bytes(8, 3, 5) == 3 * 2 * 4 and lib.qnn_packed_bytes(0, 3, 5) == 60
    assert lib.qnn_packed_bytes(7, 3, 5) == 0
    expect(lib.qnn_pack_f32(None, FAKE, 4, 8, 1, 1, 1, None), EINVAL, "pack null")
    expect(lib.qnn_pack_f32(FAKE, FAKE, 4, 0, 1, 1, 1, None), EINVAL, "pack channels=0")
    expect(lib.qnn_pack_f32(FAKE, FAKE, 4, 8, 3, 1, 4, None), EINVAL, "pack fn=ternary")
    expect(lib.qnn_pack_f32(FAKE, FAKE, 4, 8, 2, 4, 1, None), EINVAL, "pack quantized into BIN")
    expect(lib.qnn_pack_f32(FAKE, FAKE, 4, 8, 2, 8, 4, None), EINVAL, "pack nb=8 into I4")
    expect(lib.qnn_pack_f32(FAKE, FAKE, 4, 8, 2, 4, 5, None), EINVAL, "pack store=5")
    expect(lib.qnn_pack_f32(FAKE, FAKE, 0, 8, 2, 4, 4, None), 0, "pack pixels=0")
    expect(lib.qnn_unpack_f32(None, FAKE, 4, 8, 4, 4, None), EINVAL, "unpack null")
    # prepack: argument checks
    out = vp(0)
    P = lambda *a: lib.qnn_prepack_weights(*a, ctypes.byref(out))
    expect(P(2, 4, 1.0, None, 3, 3, 16, 16, None, 1, 1, 4, None), EINVAL, "prepack null kernel")
    expect(lib.qnn_prepack_weights(2, 4, 1.0, FAKE, 3, 3, 16, 16, None, 1, 1, 4, None, None), EINVAL, "prepack null out")
    expect(P(2, 4, 1.0, FAKE, 0, 3, 16, 16, None, 1, 1, 4, None), EINVAL, "prepack kh=0")
    expect(P(2, 4, 1.0, FAKE, 3, 3, -1, 16, None, 1, 1, 4, None), EINVAL, "prepack cin<0")
    expect(P(2, 4, 1.0, FAKE, 5, 5, 16, 16, None, 1, 1, 4, None), EUNSUP, "prepack 5x5")
    expect(P(2, 4, 1.0, FAKE, 3, 3, 16, 16, None, 0, 1, 4, None), EINVAL, "prepack stride=0")
    expect(P(9, 4, 1.0, FAKE, 3, 3, 16, 16, None, 1, 1, 4, None), EINVAL, "prepack wkind=9")
    expect(P(2, 1, 1.0, FAKE, 3, 3, 16, 16, None, 1, 1, 4, None), EINVAL, "prepack wbits=1")
    expect(P(2, 30, 1.0, FAKE, 3, 3, 16, 16, None, 1, 1, 0, None), EINVAL, "prepack wbits=30")
    expect(P(2, 4, 0.0, FAKE, 3, 3, 16, 16, None, 1, 1, 4, None), EINVAL, "prepack H=0")
    expect(P(2, 4, 1.0, FAKE, 3, 3, 16, 16, None, 1, 1, 1, None), EINVAL, "prepack quantized into BIN")
    expect(P(2, 8, 1.0, FAKE, 3, 3, 16, 16, None, 1, 1, 4, None), EINVAL, "prepack 8 bits into I4")
    expect(P(1, 1, 0.5, FAKE, 3, 3, 16, 16, None, 1, 1, 1, None), EINVAL, "prepack binary H!=1 packed")
    expect(P(2, 4, 1.0, FAKE, 3, 3, 16, 16, None, 1, 1, 3, None), EINVAL, "prepack store=3")
    assert not out.value
    # prepack: well-formed request, no device -> the HIP runtime refuses; everything allocated so far is released
    for args in ((2, 4, 1.0, FAKE, 3, 3, 64, 64, FAKE, 1, 1, 4, None), (1, 1, 1.0, FAKE, 3, 3, 64, 64, None, 1, 1, 1, None),
                 (3, 1, 1.0, FAKE, 1, 1, 64, 10, FAKE, 1, 0, 8, None), (0, 1, 1.0, FAKE, 3, 3, 3, 16, None, 2, 1, 0, None)):
        expect(P(*args), (EHIP, ENOMEM), "prepack without a device")
        assert not out.value
    expect(lib.qnn_free_weights(None), 0, "free(NULL)")
    expect(lib.qnn_weights_dequant(None, FAKE, None), EINVAL, "dequant null handle")
    # forward entry points: null handle / pointers
    expect(lib.qnn_conv2d_forward(None, FAKE, 4, 4, 1, 8, 8, None, FAKE, None), EINVAL, "conv null handle")
    expect(lib.qnn_dense_forward(None, FAKE, 4, 4, 1, None, FAKE, None), EINVAL, "dense null handle")
    expect(lib.qnn_conv2d_forward_f32in(None, FAKE, 1, 1, 1, 8, 8, None, FAKE, None, 0, None), EINVAL, "f32in null handle")
    assert lib.qnn_conv2d_workspace_bytes(None, 1, 8, 8) == 0
    # the fold entry points (ABI 4): null handles / pointers
    lib.qnn_fold_prepare.argtypes = [vp, ci, ci, vp, vp, ctypes.POINTER(vp)]
    lib.qnn_fold_free.argtypes = [vp]
    lib.qnn_fold_info.argtypes = [vp, vp]
    lib.qnn_fold_constants.argtypes = [vp, vp, vp, vp, vp]
    lib.qnn_fold_eval.argtypes = [vp, ci, vp, vp, vp, sz, vp]
    fo = vp(0)
    expect(lib.qnn_fold_prepare(None, 4, 4, FAKE, None, ctypes.byref(fo)), EINVAL, "fold_prepare null weights")
    expect(lib.qnn_fold_prepare(FAKE, 4, 4, None, None, ctypes.byref(fo)), EINVAL, "fold_prepare null epilogue")
    expect(lib.qnn_fold_prepare(FAKE, 4, 4, FAKE, None, None), EINVAL, "fold_prepare null out")
    assert not fo.value
    expect(lib.qnn_fold_free(None), 0, "fold_free(NULL)")
    expect(lib.qnn_fold_info(None, FAKE), EINVAL, "fold_info null handle")
    expect(lib.qnn_fold_constants(None, FAKE, FAKE, None, None), EINVAL, "fold_constants null handle")
    expect(lib.qnn_fold_eval(None, 0, FAKE, None, FAKE, 4, None), EINVAL, "fold_eval null handle")
    assert lib.qnn_last_kernel() is not None
    print("sanitized ABI checks passed:", n[0])
''')


def _build():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_sanitized
    return build_sanitized.build(), build_sanitized.asan_runtime()


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_argument_checks_and_error_paths_under_asan_ubsan():
    lib, rt = _build()
    assert rt, "libclang_rt.asan-x86_64.so not found next to hipcc's clang"
    env = dict(os.environ, QNN_LIB=lib, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=99",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98")
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, "rc=%d\n%s\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-6000:])
    assert "sanitized ABI checks passed" in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[-4000:]
