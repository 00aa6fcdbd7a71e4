"""The matrix-pipe ceiling of an x3 convolution on THIS box (tools/dev/x3_ceiling_probe.hip): table for profiles/r06_x3_ceiling_probe.txt.

    python tools/dev/x3_ceiling.py [iters]

x3-equivalent TFLOP/s = raw MFMA rate / 3 (every fp32-grade product costs three fp16 MFMAs): the number a kernel's ALGORITHMIC rate can
be compared with.  Nominal: 2500 TFLOP/s raw = 833 x3-equivalent.
"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = ((0, 'bare v_mfma_f32_32x32x16_f16 loop, random operands in registers'),
            (1, 'x3 product loop, operands from LDS (8 ds_read_b128 per 12 MFMAs), nothing else'),
            (2, 'the same + 2 scalar-f32 VALU per MFMA gap (conversion arithmetic, no memory side)'),
            (3, 'MX-fp8 lever: a_hi.b_hi on fp16 + the two low terms on fp8 K=64 MFMAs (raw = x3-equivalent issues)'))


def load():
    lib = ctypes.CDLL(os.path.join(HERE, 'libx3_ceiling_probe.so'))
    lib.x3_probe_run.restype = ctypes.c_int
    lib.x3_probe_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    return lib


def run(lib, variant, blocks_per_cu, iters):
    res = (ctypes.c_double * 4)()
    rc = lib.x3_probe_run(variant, blocks_per_cu, iters, res)
    if rc != 0:
        raise RuntimeError(f'x3_probe_run({variant}, {blocks_per_cu}) -> {rc}')
    return {'ms': res[0], 'raw_tflops': res[1], 'x3_tflops': res[1] / 3.0, 'clock_ghz': res[2], 'cycles_per_mfma_simd': res[3]}


def practical_peak(iters=4000):
    """x3-equivalent TFLOP/s of variant 1 at two blocks per CU (the halo kernels' occupancy): the loop of the dominant kernel with everything but
    its MFMAs and fragment reads removed.  Best of three (the clock follows the moment)."""
    lib = load()
    runs = [run(lib, 1, 2, iters) for _ in range(3)]
    best = max(runs, key=lambda r: r['raw_tflops'])
    bare = max((run(lib, 0, 2, iters) for _ in range(2)), key=lambda r: r['raw_tflops'])
    return best, bare


if __name__ == '__main__':
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    lib = load()
    print('# tools/dev/x3_ceiling.py %d   (1 x MI355X; 256 threads per block; raw = MFMA FLOPs executed, x3 = raw / 3 = fp32-grade products; '
          'clock = s_memtime / s_memrealtime inside the kernel; cycles = shader cycles per MFMA and SIMD, 16 = back to back)' % iters)
    print('%-104s %9s %8s %9s %9s %7s %7s %9s' % ('variant', 'blocks/CU', 'ms', 'raw TF/s', 'x3 TF/s', 'of 833', 'GHz', 'cyc/MFMA'))
    for v, name in VARIANTS:
        for bpc in (1, 2):
            for rep in range(2):
                it = iters // 4 if v == 3 else iters
                r = run(lib, v, bpc, it)
                print('%-104s %9d %8.2f %9.1f %9.1f %7.3f %7.3f %9.2f' % (name, bpc, r['ms'], r['raw_tflops'], r['x3_tflops'], r['x3_tflops'] / (2500.0 / 3.0),
                                                                      r['clock_ghz'], r['cycles_per_mfma_simd']))
