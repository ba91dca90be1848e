"""CPU test of the block kernel's own log / sin / cos (neuralmelting_amd/csrc/nm_math.h, used by the Box-Muller draw of
`velocity create`): the header is plain C++, compiled here for the host and compared with libm over random arguments of the
ranges the kernel uses (log of 1 - u in [2^-53, 1], sin and cos of 2 pi u).  The oracle calls libm for the same draws, so this
bounds the difference between the two paths at its source: within 1 ulp."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_math_is_within_one_ulp_of_libm(tmp_path):
    exe = str(tmp_path / 'nm_math_check')
    subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-I', os.path.join(ROOT, 'neuralmelting_amd', 'csrc'), '-o', exe,
                           os.path.join(ROOT, 'scripts', 'nm_math_check.cpp')])
    r = subprocess.run([exe, '3000000'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r'log ([0-9.]+)\s+sin ([0-9.]+)\s+cos ([0-9.]+)', r.stdout)
    assert m and all(float(v) <= 1.0 for v in m.groups()), r.stdout
