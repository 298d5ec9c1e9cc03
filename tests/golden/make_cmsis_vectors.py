"""Makes tests/golden/cmsis_biquad_vectors.npz and cmsis_cfft128_vectors.npz: outputs of the REFERENCE'S OWN BINARIES of CMSIS-DSP V1.4.5
`arm_cfft_f32` (128 points: arm_cfft_f32.o, arm_cfft_radix8_f32.o, arm_bitreversal2.o and their tables, linked by tests/thumb_emu.py) and
`arm_biquad_cascade_df1_f32` (member arm_biquad_cascade_df1_f32.o of the reference's
`ARM_MATH UPDATE/TeensyduinoArmMathUpdate/libarm_cortexM4lf_math.a`), executed instruction by instruction by tests/thumb_emu.py
in this container (no ARM hardware or toolchain here).  The fixture holds data only: inputs (coefficients, state, samples) and the
bit patterns the object produced, plus the sha256 of the executed code section.  Run from the repository root:
    python tests/golden/make_cmsis_vectors.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_cmsis_object as C   # noqa: E402  (the case list and the call harness are shared with the test)
from tests import thumb_emu as T           # noqa: E402


def main():
    code, _ = T.load_function(C.ARCHIVE, "arm_biquad_cascade_df1_f32.o", "arm_biquad_cascade_df1_f32")
    out = {"object_sha256": np.array(hashlib.sha256(code).hexdigest())}
    names = []
    for i, (name, co, st, x) in enumerate(C.cases()):
        y, st_after, _ = C.run_object(code, co, st, x, in_place=True)
        names.append(name)
        out["coefs_%d" % i] = np.asarray(co, np.float32)
        out["state_%d" % i] = np.asarray(st, np.float32)
        out["x_%d" % i] = np.asarray(x, np.float32)
        out["y_bits_%d" % i] = y.view(np.uint32)
        out["state_after_bits_%d" % i] = st_after.view(np.uint32)
    out["names"] = np.array(names)
    np.savez_compressed(C.GOLDEN, **out)
    print("wrote %s: %d cases" % (C.GOLDEN, len(names)))
    # arm_cfft_f32 (128 points, forward, bit-reversed to natural order): the linked objects executed on the emulator
    cpu, img = C.cfft_image()
    out, names = {}, []
    for i, (name, x) in enumerate(C.cfft_cases()):
        names.append(name)
        out["x_%d" % i] = np.asarray(x, np.float32)
        out["y_bits_%d" % i] = C.run_cfft128(cpu, img, x).view(np.uint32)
    out["names"] = np.array(names)
    np.savez_compressed(C.GOLDEN_CFFT, **out)
    print("wrote %s: %d cases" % (C.GOLDEN_CFFT, len(names)))


if __name__ == "__main__":
    main()
