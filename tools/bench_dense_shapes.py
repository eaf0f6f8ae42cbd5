#!/usr/bin/env python3
"""Ad-hoc sweep of dense-conv shapes: python tools/bench_dense_shapes.py 'cin,cout,tin,stride[,batch]' ..."""
import pathlib, statistics, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from nb_asr_amd import hip
DEV = 'cuda:0'
for spec in sys.argv[1:]:
    v = [int(t) for t in spec.split(',')]
    cin, cout, tin, s = v[:4]
    b = v[4] if len(v) > 4 else 64
    k = v[5] if len(v) > 5 else 8
    tout = (tin + s - 1) // s
    x = torch.randn(b, cin, hip.round_up4(tin), device=DEV)
    w = torch.randn(cout, cin, k, device=DEV) * 0.02
    bias = torch.randn(cout, device=DEV)
    y = torch.empty(b, cout, hip.round_up4(tout), device=DEV)
    fn = lambda: hip.dense_conv1d_fused(x, tin, w, bias, (), y, s)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    med = statistics.median(ts)
    print(f'{spec:28s} {med * 1e3:9.1f} us  {2.0 * b * tout * cout * cin * k / med / 1e9:7.1f} TFLOP/s')
