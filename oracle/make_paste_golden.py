"""Regression vectors of the paste-back restatement (oracle/paste_oracle.py) on the synthetic 1080p / 3-face case:
tests/golden/paste_1080p_3faces.npz (sha1 of the composited frame, an 8x-decimated copy, mask statistics).

NOT generated from the reference: its arithmetic is cv2's and cv2 is not installed in the build image (parity unpinned, see the
header of paste_oracle.py).  The file pins the RESTATEMENT against accidental change; the GPU test compares the HIP kernels with
the restatement itself, bit for bit.      python oracle/make_paste_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from __graft_entry__ import load_package  # noqa: E402

load_package()
import paste_oracle as P  # noqa: E402
from comfyui_keep_amd.engine import synth  # noqa: E402

frame, faces, mats, classes = synth.synth_paste_case()
out = P.paste_faces(frame, list(faces), list(mats), list(classes))
soft0 = P.parse_soft_mask(classes[0])
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'paste_1080p_3faces.npz'),
                    sha1=np.frombuffer(hashlib.sha1(out.tobytes()).digest(), np.uint8), decimated=out[::8, ::8].copy(),
                    changed_pixels=np.int64((out != frame).any(-1).sum()), soft0_sum=np.float64(soft0.astype(np.float64).sum()),
                    soft0_center=np.float32(soft0[256, 256]))
print('changed pixels', int((out != frame).any(-1).sum()), 'sha1', hashlib.sha1(out.tobytes()).hexdigest())
