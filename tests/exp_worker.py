"""Child process of tests/test_gpu_experiments.py: analyses one seeded track with whatever library and environment the
parent selected (HEADROOM_HIP_LIB, HR_*), prints the result as JSON."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import headroom_amd as hr  # noqa: E402
import signals  # noqa: E402

fs = int(sys.argv[1])
pcm = signals.seeded_noise(31 + fs, fs * 9 + 321, 2, -12.0)
pcm[:fs] *= 0.003
with hr.Meter(device=0) as m:
    print(json.dumps(m.analyze(pcm, fs)))
