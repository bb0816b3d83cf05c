import os, sys
sys.path.insert(0, os.getcwd())
os.environ["BOD_TRACE_OPS"] = "2"
import numpy as np
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
B = 64
eng = Engine(make_config((512, 512), batch=B, mc_samples=10))
eng.load_weights(synthetic.make_weights())
eng.upload_images(synthetic.make_frames(B, 512, 512))
for i in range(3):
    eng.forward(None); eng.synchronize()
