"""Per-kernel resource table of the gfx950 build: registers, LDS, scratch, occupancy-relevant limits.

    python scripts/kernel_resources.py [--json]

Compiles every st-p3_amd/csrc/*.hip to gfx950 assembly (hipcc -S --cuda-device-only, no GPU needed) and reads the
kernel descriptors' metadata.  Things a CPU execution of the kernels cannot see and this can: register spills
(scratch), static LDS against the 64 KB default launch limit / the 160 KB of a CU, VGPR counts against the 512-register
budget (waves per SIMD = floor(512 / VGPRs))."""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt',
         '-fno-fast-math', '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', '-w']


def demangle(names):
    try:
        out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines()
        return [re.sub(r'\(anonymous namespace\)::', '', o).split('(')[0] for o in out]
    except OSError:
        return names


def kernels_of(path):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'k.s')
        subprocess.check_call([HIPCC] + FLAGS + [path, '-o', asm], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = []
    # the amdhsa metadata block lists one YAML mapping per kernel
    for block in re.findall(r'^  - \.agpr_count:.*?(?=^  - \.agpr_count:|^amdhsa\.target|\Z)', text, re.S | re.M):
        def field(name, default=0):
            m = re.search(r'\.' + name + r':\s+(\S+)', block)
            return m.group(1) if m else default
        out.append({'symbol': field('name', '?'), 'vgpr': int(field('vgpr_count')), 'agpr': int(field('agpr_count')),
                    'sgpr': int(field('sgpr_count')), 'lds_static': int(field('group_segment_fixed_size')),
                    'scratch': int(field('private_segment_fixed_size')), 'max_threads': int(field('max_flat_workgroup_size')),
                    'vgpr_spills': int(field('vgpr_spill_count')), 'sgpr_spills': int(field('sgpr_spill_count'))})
    names = demangle([k['symbol'] for k in out])
    for k, n in zip(out, names):
        k['kernel'] = n
    return out


def main():
    rows = []
    for path in sorted(glob.glob(os.path.join(ROOT, 'st-p3_amd', 'csrc', '*.hip'))):
        for k in kernels_of(path):
            k['file'] = os.path.basename(path)
            rows.append(k)
    if '--json' in sys.argv:
        json.dump(rows, sys.stdout, indent=1)
        return
    print(f'{"file":16s} {"kernel":46s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"lds":>7s} {"scratch":>8s} {"spills":>7s} {"waves/SIMD":>10s}')
    for k in rows:
        regs = max(k['vgpr'] + k['agpr'], 1)
        print(f'{k["file"]:16s} {k["kernel"][:46]:46s} {k["vgpr"]:5d} {k["agpr"]:5d} {k["sgpr"]:5d} {k["lds_static"]:7d} '
              f'{k["scratch"]:8d} {k["vgpr_spills"] + k["sgpr_spills"]:7d} {min(8, 512 // regs):10d}')


if __name__ == '__main__':
    main()
