import sys, numpy as np
sys.path.insert(0, '/root/repo')
from bayes_od_rc_amd.engine import stage_conv_wgrad
rng = np.random.default_rng(0)
for (b, h, w, cin, cout, k) in ((3, 90, 160, 256, 256, 3), (3, 180, 320, 64, 64, 3), (3, 45, 80, 256, 1024, 1), (8, 64, 64, 256, 256, 3)):
    x = rng.normal(0, 1, (b, h, w, cin)).astype(np.float32)
    dy = rng.normal(0, 1, (b, h, w, cout)).astype(np.float32)
    for _ in range(2):
        stage_conv_wgrad(x, dy, (k, k))
