"""Winograd F(2x2,3x3) vs direct implicit-GEMM on the path's 3x3 stride-1 shapes: correctness (vs f64) + time."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-faster-rcnn_amd")]
import numpy as np, torch
from frcnn_hip import ops, lib
dev = torch.device("cuda:0")
shapes = {"b1c2x4": (4,150,250,64,64), "b2c2x4": (4,75,125,128,128), "b3c2x4": (4,38,63,256,256), "rpnx4": (4,38,63,1024,512),
          "b4c2": (300,7,7,512,512), "b4c2x4": (1200,7,7,512,512), "b3c2": (1,38,63,256,256), "rpn": (1,38,63,1024,512)}
def timeit(fn, n=8, rounds=5):
    ts = []
    for r in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1) * 1000 / n)
    return float(np.median(ts))
MM = int(sys.argv[1]) if len(sys.argv) > 1 else 2
GG = ops.winograd_points(MM)
print("F(%d,3) %-7s %9s %9s %9s %9s %9s | %9s %9s" % (MM, "shape", "direct", "wino", "in_tr", "gemm", "out_tr", "err_dir", "err_wino"))
for name, (N,H,W,Cin,Cout) in shapes.items():
    if MM == 7 and H != 7: continue
    torch.manual_seed(0)
    x = torch.randn(N,H,W,Cin, device=dev).relu(); w_hwio = (torch.randn(3,3,Cin,Cout) * (2.0/(9*Cin))**0.5).numpy(); b = torch.randn(Cout, device=dev)
    wp = torch.from_numpy(np.ascontiguousarray(w_hwio.transpose(3,0,1,2))).to(dev)
    u = torch.from_numpy(ops.winograd_filter_transform(w_hwio, None, MM)).to(dev)
    out_d = torch.empty(N,H,W,Cout, device=dev); out_w = torch.empty(N,H,W,Cout, device=dev)
    T = ops.winograd_tiles(N,H,W,MM)
    v = torch.empty(GG,T,Cin, device=dev); m = torch.empty(GG,T,Cout, device=dev)
    fd = lambda: ops.conv2d(x, wp, b, 3, 3, 1, (1,1,1,1), 1, None, 1, out=out_d)
    fw = lambda: ops.conv3x3_winograd(x, u, b, 1, out=out_w, v_buf=v, m_buf=m)
    fd(); fw(); torch.cuda.synchronize()
    nref = min(N, 8)
    ref = torch.nn.functional.conv2d(x[:nref].double().permute(0,3,1,2), torch.from_numpy(w_hwio).to(dev).double().permute(3,2,0,1), b.double(), padding=1).relu().permute(0,2,3,1)
    sc = ref.abs().max().item()
    ed = (out_d[:nref].double()-ref).abs().max().item()/sc; ew = (out_w[:nref].double()-ref).abs().max().item()/sc
    L = lib(); S = torch.cuda.current_stream().cuda_stream
    t_d = timeit(fd); t_w = timeit(fw)
    t_i = timeit(lambda: ops.winograd_input_transform(x, v, MM))
    t_g = timeit(lambda: ops.call("frcnn_gemm_batched_nt", v.data_ptr(), u.data_ptr(), m.data_ptr(), GG, T, Cout, Cin, S))
    t_o = timeit(lambda: ops.winograd_output_transform(m, b, 1, out_w, MM))
    print("%-7s %9.1f %9.1f %9.1f %9.1f %9.1f | %9.2e %9.2e" % (name, t_d, t_w, t_i, t_g, t_o, ed, ew))
