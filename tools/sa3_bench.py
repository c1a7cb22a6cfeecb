#!/usr/bin/env python
"""Times lt_softargmax3d_fwd on the BASELINE shape (B x 64^3 x 17 channels-last fp32 logits), hipEvents around 5 launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learnable-triangulation-pytorch_amd"))
import torch

import lt_hip as H

B, J, V = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 17, 64
lib = H.lib()
dev = "cuda:0"
lg = torch.randn(B, V ** 3, J, device=dev)
cv = torch.randn(B, V ** 3, 3, device=dev)
kp = torch.empty(B, J, 3, device=dev)
pr = torch.empty(B, J, V ** 3, device=dev)
ws = torch.empty(lib.lt_softargmax3d_workspace(B, J, V ** 3), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for with_probs in (True, False):
    for _ in range(2):
        H.check(lib.lt_softargmax3d_fwd(lg.data_ptr(), cv.data_ptr(), 1.0, 1, 1, J, kp.data_ptr(), pr.data_ptr() if with_probs else None, B, J, V ** 3, ws.data_ptr(), st), "sa3")
    e0, e1 = H.Event(), H.Event()
    e0.record(st)
    for _ in range(5):
        H.check(lib.lt_softargmax3d_fwd(lg.data_ptr(), cv.data_ptr(), 1.0, 1, 1, J, kp.data_ptr(), pr.data_ptr() if with_probs else None, B, J, V ** 3, ws.data_ptr(), st), "sa3")
    e1.record(st)
    print("softargmax3d B=%d probs=%d: %.0f us" % (B, with_probs, e0.elapsed_ms(e1) / 5 * 1e3))
