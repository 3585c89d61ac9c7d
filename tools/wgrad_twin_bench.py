"""Weight-gradient kernel fed from fp32 operands vs from bf16 twins, per shape (HIP events, cold-ish: rotating buffers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CDETR_TUNING", "1")
import torch
from counting_detr_amd import ops
ops.PRECISION, ops.PRECISION_BWD = 1, 3
dev = "cuda"
shapes = [(5000, 256, 1024, 1), (5000, 1024, 256, 1), (5000, 512, 512, 9), (20000, 128, 128, 9), (20000, 512, 128, 1), (5000, 2048, 512, 1), (5000, 256, 256, 9)]
NB = 6
for (P, Nout, Cin, taps) in shapes:
    k = 3 if taps == 9 else 1
    hw = int(round((P // 2) ** 0.5))
    res = {}
    for twins in (False, 1, 2, 4):                 # fp32 operands | bf16 twins with KP = 1 / 2 / 4 32-pixel blocks per barrier (CDETR_WGRAD_KP)
        os.environ["CDETR_WGRAD_KP"] = str(int(twins) or 1)
        dz = [torch.randn(2, hw, hw, Nout, device=dev) for _ in range(NB)]
        x = [torch.randn(2, hw, hw, Cin, device=dev) for _ in range(NB)]
        dz16 = [t.to(torch.bfloat16) for t in dz]
        x16 = [t.to(torch.bfloat16) for t in x]
        w = torch.nn.Parameter(torch.zeros(Nout, Cin, k, k, device=dev).contiguous(memory_format=torch.channels_last))
        w.grad = torch.zeros_like(w)
        def run(i):
            j = i % NB
            if twins:
                ops.conv_wgrad_(dz[j], x[j], w, None, pad=k // 2, dz16=dz16[j], x16=x16[j])
            else:
                ops.conv_wgrad_(dz[j], x[j], w, None, pad=k // 2)
        for i in range(6):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        res[twins] = e0.elapsed_time(e1) / 30 * 1e3
    fl = 2.0 * (2 * hw * hw) * Nout * Cin * taps
    print(f"P={2*hw*hw} Nout={Nout} Cin={Cin} taps={taps}: fp32 operands {res[False]:.1f} us ({fl/res[False]/1e6:.0f} TF)   bf16 twins KP=1 {res[1]:.1f} us ({fl/res[1]/1e6:.0f} TF)"
          f"   KP=2 {res[2]:.1f} us ({fl/res[2]/1e6:.0f} TF)   KP=4 {res[4]:.1f} us ({fl/res[4]/1e6:.0f} TF)")
