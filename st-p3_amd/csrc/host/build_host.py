"""Builds st-p3_amd/stp3_amd/_stp3_host*.so (the experimental C++ launch path) with plain g++ against the torch
headers.  No device code: libstp3hip.so is dlopen()ed at run time.  Usage: python build_host.py"""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension as ce

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..', '..'))
OUT = os.path.join(ROOT, 'st-p3_amd', 'stp3_amd', '_stp3_host.so')
SRC = os.path.join(HERE, 'stp3_host.cpp')


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(
            os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, 'include', 'stp3_hip.h'))):
        return OUT
    inc = ce.include_paths() + [sysconfig.get_paths()['include'], '/opt/rocm/include', os.path.join(ROOT, 'include')]
    lib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = ['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-DTORCH_EXTENSION_NAME=_stp3_host', '-D__HIP_PLATFORM_AMD__=1',
           '-DUSE_ROCM=1', f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}', '-Wno-deprecated-declarations']
    cmd += [f'-I{p}' for p in inc]
    cmd += [SRC, '-o', OUT, f'-L{lib}', '-ltorch', '-ltorch_cpu', '-ltorch_python', '-lc10', '-lc10_hip', '-ldl',
            f'-Wl,-rpath,{lib}']
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
