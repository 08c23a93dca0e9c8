"""Shape fuzzer for the HOST half of the C-ABI (argument checks, workspace / bank sizing, split-K cost models, grid arithmetic)
running against a build of libmaua_hip whose host code is instrumented with AddressSanitizer + UndefinedBehaviorSanitizer
("sanitizers": device-side ASan is not available on this pool, so the boundary is sanitised on the CPU).

    python maua-style_amd/build_native.py --sanitize        # -> maua-style_amd/csrc/build/libmaua_hip_san.so
    LD_PRELOAD=<clang_rt.asan> python tools/fuzz_abi_host.py LIB [N] [SEED]

No GPU is needed or used: pointers are fake non-null addresses that host code only passes on; a launch on a GPU-less box comes
back as a HIP error code, which is a legal return value here.  What must hold: no sanitizer report (the process would abort),
every size function returns a value that is consistent with its arguments, invalid arguments give a negative return code."""
import ctypes
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "maua-style_amd")]


def main():
    lib_path = sys.argv[1]
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    os.environ["MAUA_HIP_LIB"] = lib_path
    import hip  # binds every symbol of include/maua_hip.h with its argtypes (no torch tensors are created here)
    L = hip.lib()
    P = 0x10000  # fake device pointer
    edge = [-7, -1, 0, 1, 2, 3, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 255, 256, 384, 511, 512, 513, 1000,
            1024, 2048, 4096, 65535, 65536, 1 << 20, (1 << 31) - 1]

    def dim(big=False):
        v = rng.choice(edge) if rng.random() < 0.7 else rng.randint(-4, 5000)
        return v if big or v < 100000 else rng.choice([64, 512, 1024])

    checked = 0
    for case in range(n_cases):
        big = rng.random() < 0.15  # a share of cases with dimensions up to INT_MAX: the checks must come before the arithmetic
        n, cin, cout, h, w = dim(big), dim(big), dim(big), dim(big), dim(big)
        n = n if rng.random() < 0.2 else rng.choice([1, 1, 1, 2, 3, 4, 16])
        k, stride, pad = rng.choice([1, 3, 5, 11, 2, 0, -1]), rng.choice([1, 1, 1, 2, 4, 0]), rng.choice([0, 1, 2, 3, 5, -1])
        L.maua_set_split_batch_hint(rng.choice([1, 1, 4, 16, 0, -3]))
        # size functions: any int arguments at all
        sizes = [L.maua_conv_workspace_bytes(n, cin, h, w, cout, k, k, stride, pad),
                 L.maua_conv_x6_workspace_bytes(n, cin, h, w, cout, pad), L.maua_conv_x3_workspace_bytes(n, cin, h, w, cout, pad),
                 L.maua_conv_x3w_workspace_bytes(n, cin, h, w, cout, pad), L.maua_conv_kxk_x3_workspace_bytes(n, cin, h, w, cout, k, pad),
                 L.maua_conv1x1_x3_workspace_bytes(n, cin, h * w if abs(h * w) < 1 << 40 else 1, cout),
                 L.maua_conv_x6_bank_bytes(cout, cin), L.maua_conv_x3_bank_bytes(cout, cin), L.maua_conv_x3w_bank_bytes(cout, cin),
                 L.maua_conv_kxk_x3_bank_bytes(cout, cin, k), L.maua_gram_workspace_bytes(cin, h * w if abs(h * w) < 1 << 40 else 1),
                 L.maua_reduce_workspace_bytes(h * w), L.maua_lbfgs_state_bytes(h * w, rng.choice([1, 5, 100, 254, 255, 0, -1])),
                 L.maua_channel_stats_workspace_bytes(h, w), L.maua_loss_ledger_bytes(n, cin),
                 L.maua_conv_image_bank_bytes(cout, cin)]
        assert all(s >= 0 for s in sizes)
        if min(n, cin, cout, h, w) <= 0:
            assert sizes[1] == 0 and sizes[2] == 0 and sizes[3] == 0, (n, cin, cout, h, w, sizes)
        if min(cin, cout) <= 0:
            assert sizes[6] == 0 and sizes[7] == 0 and sizes[8] == 0 and sizes[15] == 0, (cin, cout, sizes)
        L.maua_pool_out_size(h, rng.choice([2, 3, 0, -1]), rng.choice([2, 1, 0]), rng.randint(0, 1))
        L.maua_conv_x3w_supported(cin, h, w, pad)
        assert L.maua_conv_x3w_split(n, cin, h, w, cout, pad) >= 0 and L.maua_conv_x3w_dmat_bank_bytes(cin) >= 0
        L.maua_conv_pack_dmat_x3w(None, cin, None, None, None)
        L.maua_conv_pack_filters_image(None, None, None, cout, cin, None)
        # compute entry points: null pointers and bad dims must be refused with a negative code before any launch; good
        # arguments reach the launch (a HIP error on this GPU-less box, or 0 on a GPU box where P would fault - so only
        # argument sets that fail validation use P there)
        no_gpu = os.environ.get("MAUA_FUZZ_ASSUME_NO_GPU", "1") == "1"
        ws_bytes = rng.choice([0, 1 << 10, 1 << 30])
        ptr = P if no_gpu else None
        rcs = [
            L.maua_conv3x3_x3w(ptr, ptr, 1.0, ptr, None, ptr, n, cin, h, w, cout, pad, 1, 0, ptr, ws_bytes, None),
            L.maua_conv3x3_x3w_relu_pool(ptr, ptr, 1.0, ptr, ptr, ptr, n, cin, h, w, cout, pad, ptr, ws_bytes, None),
            L.maua_conv3x3_x3w_gram(ptr, ptr, 1.0, ptr, ptr, ptr, ptr, n, cin, h, w, cout, pad, 0, ptr, ws_bytes, None),
            L.maua_conv3x3_x3w_unpool(ptr, ptr, rng.randint(0, 1), ptr, 1.0, rng.choice([None, ptr]), rng.choice([None, ptr]), ptr, ptr, n, cin, h, w,
                                      cout, pad, ptr, ws_bytes, None),
            L.maua_conv3x3_x3(ptr, ptr, 1.0, ptr, None, ptr, n, cin, h, w, cout, pad, 1, 0, ptr, ws_bytes, None),
            L.maua_conv3x3_x6(ptr, ptr, ptr, None, ptr, n, cin, h, w, cout, pad, 1, 0, ptr, ws_bytes, None),
            L.maua_conv3x3_image(ptr, ptr, ptr, n, cin, h, w, cout, pad, 1, None),
            L.maua_conv2d_fwd(ptr, None, ptr, ptr, ptr, n, cin, h, w, cout, k, k, stride, pad, 1, 0, ptr, ws_bytes, None),
            L.maua_conv2d_bwd_data(ptr, None, ptr, ptr, None, ptr, n, cin, h, w, cout, k, k, stride, pad, 0, ptr, ws_bytes, None),
            L.maua_conv_kxk_x3(ptr, ptr, 1.0, ptr, None, ptr, n, cin, h, w, cout, k, pad, 1, 0, ptr, ws_bytes, None),
            L.maua_conv1x1_x3(ptr, None, ptr, ptr, None, ptr, n, cin, h * w if abs(h * w) < 1 << 40 else 1, cout, 1, 0, ptr, ws_bytes, None),
            L.maua_pool2d_fwd(ptr, ptr, n, cin, h, w, rng.choice([2, 3]), 2, rng.randint(0, 1), rng.randint(0, 1), None),
            L.maua_pool2x2_fwd_codes(ptr, ptr, ptr, n, cin, h, w, None),
            L.maua_pool2x2_bwd_codes(ptr, ptr, ptr, n, cin, h, w, rng.randint(0, 1), None),
            L.maua_gram_fwd(ptr, ptr, None, cin, h * w if abs(h * w) < 1 << 40 else 1, 1.0, 0, ptr, ws_bytes, None),
            L.maua_gram_bwd(ptr, ptr, None, None, ptr, cin, h * w if abs(h * w) < 1 << 40 else 1, 0, ptr, ws_bytes, None),
            L.maua_mse_fwd_bwd(ptr, ptr, ptr, h * w, 1.0, 1.0, 0, 0, ptr, ptr, ws_bytes, None),
            L.maua_mse_fwd_bwd_ledger(ptr, ptr, ptr, h * w, 1.0, 1.0, 0, 0, ptr, rng.choice([0, 3, -1]), None),
            L.maua_tv_fwd_bwd_ledger(ptr, ptr, n, cin, h, w, 1.0, 0, ptr, rng.choice([0, 3, -1]), None),
            L.maua_gram_fwd_mse_ledger(ptr, ptr, None, cin, h * w if abs(h * w) < 1 << 40 else 1, 1.0, 0, ptr, ptr, 1.0, 1.0, ptr,
                                       rng.choice([0, 3, -1]), ptr, ws_bytes, None),
            L.maua_loss_ledger_sum(ptr, n, cin, ptr, ptr, None),
            L.maua_loss_ledger_sum_f64(ptr, n, cin, ptr, ptr, ptr, None),
            L.maua_gram_partial(ptr, None, cin, h * w if abs(h * w) < 1 << 40 else 1, 0, ptr, ws_bytes, None),
            L.maua_gram_partial_batch(rng.choice([0, -1, 9, 1000]), None, None, None, None, None, None, None, None),
            L.maua_gram_row_means(ptr, ptr, cin, h * w if abs(h * w) < 1 << 40 else 1, ptr, ws_bytes, None),
            L.maua_conv_pack_dmat_x3w_batch(rng.choice([0, -1, 5, 1000]), None, None, None, None, None),
            L.maua_gram_finish_mse_batch(rng.choice([0, -1, 9, 1000]), None, None, None, None, None, None, None, None, None, None, None, None, None),
            L.maua_gram_mse_ledger_supported(cin),
            L.maua_tv_fwd_bwd(ptr, ptr, n, cin, h, w, 1.0, 0, ptr, ptr, ws_bytes, None),
            L.maua_resize_bilinear(ptr, ptr, n, h, w, cout, cin, 0.5, 0.5, None),
            L.maua_channel_stats(ptr, None, 1e-3, n, rng.randint(-1, 3), h, w, ptr, ptr, ws_bytes, None),
            L.maua_lbfgs_iterate(ptr, ptr, ptr, None, h * w, rng.choice([1, 100, 254, 255]), 1.0, -1.0, -1.0, None),
        ]
        if min(n, cin, cout, h, w) <= 0:  # (the 1x1 entry takes the product h * w, which two negative extents make positive)
            assert all(rc < 0 for rc in rcs[:11]), (n, cin, cout, h, w, rcs)
        checked += len(rcs) + len(sizes)
    L.maua_set_split_batch_hint(1)
    print(f"fuzz_abi_host: {n_cases} cases, {checked} calls, no sanitizer report, return codes consistent")


if __name__ == "__main__":
    main()
