"""TEST INFRASTRUCTURE -- build libstp3hip_cpu.so: the repository's .hip sources compiled for the HOST against the
stand-in HIP runtime of this directory (hip/hip_runtime.h), same C ABI as libstp3hip.so.

    python tests/hipcpu/build.py [out.so]

The only source rewrite is `extern __shared__ T name[];` -> `T* name = (T*)hipcpu::dyn_lds();` (a block-scope extern
inside an anonymous namespace cannot be defined from outside); everything else is handled by macros in the header.
Needs a clang++ that accepts ext_vector_type / __bf16 on x86: the one that ships with ROCm."""
import glob
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CLANG = os.environ.get('HIPCPU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
DEFAULT_OUT = os.path.join(HERE, 'libstp3hip_cpu.so')

EXTERN_SHARED = re.compile(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];')


def asan_runtime():
    """Path of the AddressSanitizer runtime to LD_PRELOAD when running a library built with HIPCPU_ASAN=1."""
    return subprocess.check_output([CLANG, '-print-file-name=libclang_rt.asan-x86_64.so'], text=True).strip()


def build(out=DEFAULT_OUT, sources=None, asan=False):
    sources = sources or sorted(glob.glob(os.path.join(ROOT, 'st-p3_amd', 'csrc', '*.hip')))
    newest = max(os.path.getmtime(p) for p in sources + glob.glob(os.path.join(HERE, '*.cpp')) +
                 glob.glob(os.path.join(HERE, 'hip', '*.h')) + [os.path.abspath(__file__)])
    if os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    with tempfile.TemporaryDirectory() as tmp:
        cpps = []
        for src in sources:
            text = open(src).read()
            text = EXTERN_SHARED.sub(lambda m: f'{m.group(1)}* {m.group(2)} = ({m.group(1)}*)hipcpu::dyn_lds();', text)
            dst = os.path.join(tmp, os.path.basename(src)[:-4] + '.cpp')
            with open(dst, 'w') as f:
                f.write(f'#line 1 "{src}"\n' + text)
            cpps.append(dst)
        extra = ['-fsanitize=address', '-shared-libasan', '-fno-omit-frame-pointer', '-g'] if asan else []
        cmd = [CLANG, '-std=c++17', '-O1', '-fPIC', '-shared', '-pthread', '-ffp-contract=off', '-w'] + extra + [
               '-I', HERE, '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'st-p3_amd', 'csrc'),
               os.path.join(HERE, 'hipcpu_runtime.cpp')] + cpps + ['-o', out]
        subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    print(build(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_OUT, asan=os.environ.get('HIPCPU_ASAN') == '1'))
