"""CPU test infrastructure: record what the host code hands to the C ABI.

``build_recorder(path)`` compiles (gcc) a stand-in for libstp3hip.so generated from the prototypes of
include/stp3_hip.h.  Every entry point appends one line to $STP3_TRACE_LOG -- name, the dims struct as hex, every
scalar argument, and per pointer argument ``N`` (null) / its alias class within the call, plus a checksum of the
first bytes of the INPUT buffers that hold caller data (activations, gradients, weights, BatchNorm parameters) --
and returns 0 without touching a GPU; the size queries are answered by the real library ($STP3_REAL_LIB).

tests/model_trace.py and tests/bench_dryrun.py run the whole training step / bench.py against it.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'stp3_hip.h')

# input buffers whose contents come from the caller (not from an earlier, here mocked, kernel)
DATA_PARAMS = {'x', 'dy', 'w', 'res', 'gamma', 'beta', 'bias', 'sbias', 'oscale', 'grad_out', 'seg_off', 'gate', 'add'}


def prototypes():
    src = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    out = []
    for m in re.finditer(r'\b(int|const char\*)\s+(stp3_\w+)\s*\(([^)]*)\)\s*;', src):
        params = [' '.join(p.split()) for p in m.group(3).split(',')]
        out.append((m.group(1), m.group(2), [] if params == ['void'] else params))
    return out


def recorder_source():
    lines = ['#include <dlfcn.h>', '#include <stdint.h>', '#include <stdio.h>', '#include <stdlib.h>',
             '#include <string.h>', '#include "stp3_hip.h"', '',
             'static FILE* logf(void) { static FILE* f; if (!f) f = fopen(getenv("STP3_TRACE_LOG"), "a"); return f; }',
             'static void* real(const char* name) { static void* h; if (!h) h = dlopen(getenv("STP3_REAL_LIB"), RTLD_NOW);'
             ' return dlsym(h, name); }',
             'static void hex(FILE* f, const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p;'
             ' for (size_t i = 0; i < n; ++i) fprintf(f, "%02x", b[i]); }',
             'static unsigned sum32(const void* p) { const unsigned char* b = (const unsigned char*)p; unsigned s = 0;'
             ' for (int i = 0; i < 32; ++i) s = s * 131u + b[i]; return s; }', '']
    for ret, name, params in prototypes():
        if name == 'stp3_version':
            lines.append('const char* stp3_version(void) { return "stp3hip recorder"; }')
            continue
        names = [p.split('*')[-1].split()[-1] for p in params]
        lines.append(f'{ret} {name}({", ".join(params)}) {{')
        lines.append(f'    FILE* f = logf(); fprintf(f, "{name}");')
        if name.endswith('_bytes') or name.endswith('_workspace'):      # size queries: answered by the real library
            types = ', '.join(p.rsplit(' ', 1)[0] if '*' not in p.rsplit(' ', 1)[1] else p.rsplit('*', 1)[0] + '*'
                              for p in params)
            lines.append(f'    int rc = ((int (*)({types}))real("{name}"))({", ".join(names)});')
            for p, n in zip(params[:-1], names[:-1]):
                if '_dims*' in p:
                    lines.append(f'    fprintf(f, " {n}="); hex(f, {n}, sizeof(*{n}));')
                else:
                    lines.append(f'    fprintf(f, " {n}=%lld", (long long){n});')
            lines.append(f'    fprintf(f, " -> rc=%d bytes=%zu\\n", rc, *{names[-1]}); fflush(f); return rc;')
            lines.append('}')
            continue
        ptrs = [n for p, n in zip(params, names) if '*' in p and '_dims*' not in p and n != 'stream']
        for p, n in zip(params, names):
            if '_dims*' in p:
                lines.append(f'    fprintf(f, " {n}="); hex(f, {n}, sizeof(*{n}));')
            elif n == 'stream':
                continue
            elif '*' in p:
                k = ptrs.index(n)
                alias = ' '.join(f'else if ((const void*){n} == (const void*){ptrs[j]}) fprintf(f, " {n}=A{j}");'
                                 for j in range(k))
                lines.append(f'    if (!{n}) fprintf(f, " {n}=N"); {alias} else fprintf(f, " {n}=A{k}");')
                if n in DATA_PARAMS:
                    lines.append(f'    if ({n}) fprintf(f, ":%08x", sum32({n}));')
            elif p.startswith('float') or p.startswith('double'):
                lines.append(f'    fprintf(f, " {n}=%.9g", (double){n});')
            else:
                lines.append(f'    fprintf(f, " {n}=%lld", (long long){n});')
        lines.append('    fprintf(f, "\\n"); fflush(f); return 0;')
        lines.append('}')
    return '\n'.join(lines) + '\n'


def build_recorder(path):
    c = path[:-3] + '.c'
    with open(c, 'w') as f:
        f.write(recorder_source())
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-O1', '-w', '-I', os.path.join(ROOT, 'include'), c, '-o', path,
                           '-ldl'])
    return path
