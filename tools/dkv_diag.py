"""developer check: dK / dV of the default kernel vs the parameter-table alternative vs the oracle, per key block"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import harness
from test_attention_gpu import make_desc, round_inputs, parameter_rows, DKV_RS
from metal_flash_attention_amd import GEMMOperandPrecision as P
from oracle import Network, NetworkDescriptor

R, C, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (3000, 3000, 128)
causal = len(sys.argv) > 4 and sys.argv[4] == "causal"
net = Network(NetworkDescriptor(R, C, D), seed=R + C + D + 1)
desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
run = harness.DeviceRun(desc, net, causal=causal)
a = run.execute()
with parameter_rows(DKV_RS):
    run2 = harness.DeviceRun(desc, net, causal=causal)
b = run2.execute()
print({t.name: k.variant for t, k in run.kernels.items()}, {t.name: k.variant for t, k in run2.kernels.items()})
round_inputs(net, desc)
ref = net.run(causal=causal)
for name in ("dK", "dV", "dQ"):
    ea, eb, ab = np.abs(a[name] - ref[name]), np.abs(b[name] - ref[name]), np.abs(a[name] - b[name])
    print(name, "default vs oracle %.3e  alt vs oracle %.3e  default vs alt %.3e  |ref| max %.2f" % (ea.max(), eb.max(), ab.max(), np.abs(ref[name]).max()))
    i = np.unravel_index(ab.argmax(), ab.shape)
    print("   largest default-vs-alt difference at", i, "default", a[name][i], "alt", b[name][i], "oracle", ref[name][i])
    rows = ea.max(axis=1)
    blocks = [rows[s:s + 256].max() for s in range(0, len(rows), 256)]
    print("   default vs oracle per 256-block:", " ".join("%.1e" % x for x in blocks))
