"""The C ABI from plain C (no Python in the process that steps the env): builds examples/c_abi_demo.c with gcc against
include/dpenv.h + libdpenv.so + the HIP runtime and runs it on the GPU."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_steps_the_env(tmp_path):
    exe = str(tmp_path / 'c_abi_demo')
    lib = os.path.join(ROOT, 'ml4ca_amd', 'lib')
    subprocess.check_call(['gcc', '-std=c11', '-O2', '-I', os.path.join(ROOT, 'include'), '-I', '/opt/rocm/include',
                           os.path.join(ROOT, 'examples', 'c_abi_demo.c'), '-L', lib, '-ldpenv', '-L', '/opt/rocm/lib', '-lamdhip64',
                           '-Wl,-rpath,' + lib, '-Wl,-rpath,/opt/rocm/lib', '-o', exe])
    out = subprocess.check_output([exe, '65536', '1000'], timeout=120).decode()
    m = re.search(r'= ([0-9.e+]+) env-steps/s .* mean reward (-?[0-9.]+); faults (\d+)', out)
    assert m, out
    assert float(m.group(1)) > 1e9 and int(m.group(3)) == 0 and -3.0 < float(m.group(2)) < 3.5, out
