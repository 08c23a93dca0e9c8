"""maua_conv3x3_x3w_relu_pool of two builds of the library on the same box (entry point unchanged between them).
    python tools/ab_relu_pool.py LIB_A LIB_B [image side]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "maua-style_amd"))
import torch
import hip

side = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
c_p, c_i, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
for L in libs:
    L.maua_conv3x3_x3w_relu_pool.restype = c_i
    L.maua_conv3x3_x3w_relu_pool.argtypes = [c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, ctypes.c_size_t, c_p]
torch.manual_seed(0)
for name, c, s in (("conv1_2", 64, side), ("conv2_2", 128, side // 2), ("conv3_4", 256, side // 4), ("conv4_4", 512, side // 8)):
    x = torch.relu(torch.randn(1, c, s, s, device="cuda"))
    w = torch.randn(c, c, 3, 3, device="cuda") * (2.0 / (9 * c)) ** 0.5
    b = torch.randn(c, device="cuda") * 0.1
    bf, _, wsc = hip.conv_pack_filters_x3w(w)
    pooled = torch.empty(1, c, s // 2, s // 2, device="cuda")
    codes = torch.empty(1, c, s // 2, s // 2, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    res = []
    outs = []
    for L in libs:
        def run():
            rc = L.maua_conv3x3_x3w_relu_pool(x.data_ptr(), bf.data_ptr(), float(wsc), b.data_ptr(), pooled.data_ptr(), codes.data_ptr(), 1, c, s, s, c, 1, None, 0, st)
            assert rc == 0
        best = 1e9
        for _ in range(5):
            run(); torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                run()
            e.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(e) * 100)
        res.append(best)
        outs.append(pooled.clone())
    print(f"{name} {c}ch @{s}: A {res[0]:7.1f} us   B {res[1]:7.1f} us   pooled equal {torch.equal(outs[0], outs[1])}")
