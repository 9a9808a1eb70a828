#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel the compiler emits for csrc/*.hip (hipcc -S for gfx950; no GPU needed):
`python tools/occupancy_scan.py [file.hip ...]` prints the kernels at two waves per SIMD or fewer, or with scratch.  The
compiler keeps loop-invariant LDS tables in registers and requests every independent load of an unrolled body first (round 5:
head_loss_kernel at 412 registers, a Dirichlet kernel at 372): a scan of this table after a kernel change finds that in a
minute.  tests/test_kernel_registers_cpu.py pins the budgets of the kernels that were caught."""
import concurrent.futures
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'modular_semantic_segmentation_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only']
EXTRA = {'pointwise.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}      # as the Makefile builds it


def scan_file(path):
    """[(mangled kernel name, total VGPRs incl. AGPRs, scratch bytes, waves per SIMD)] of one source file."""
    name = os.path.basename(path)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, name + '.s')
        r = subprocess.run(['hipcc'] + FLAGS + EXTRA.get(name, []) + ['-o', out, path], capture_output=True, text=True,
                           cwd=os.path.dirname(path))
        if r.returncode != 0:
            raise RuntimeError('hipcc -S %s failed:\n%s' % (name, r.stderr[-2000:]))
        rows, kern, tot, scr = [], None, None, None
        for line in open(out):
            m = re.match(r'^(_Z\S+):', line)
            if m:
                kern = m.group(1)
            elif line.startswith('; TotalNumVgprs:'):
                tot = int(line.split(':')[1])
            elif line.startswith('; ScratchSize:'):
                scr = int(line.split(':')[1])
            elif line.startswith('; Occupancy:'):
                rows.append((kern, tot, scr, int(line.split(':')[1])))
        return rows


def scan(files=None, workers=4):
    files = files or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    with concurrent.futures.ThreadPoolExecutor(workers) as ex:
        return dict(zip([os.path.basename(f) for f in files], ex.map(scan_file, files)))


def demangle(names):
    filt = shutil.which('llvm-cxxfilt') or shutil.which('c++filt')
    if not filt:
        return names
    r = subprocess.run([filt], input='\n'.join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


if __name__ == '__main__':
    if any(a in ('-h', '--help') for a in sys.argv[1:]):
        print(__doc__)
        print('  --all   list every kernel, not only the flagged ones')
        sys.exit(0)
    show_all = '--all' in sys.argv[1:]
    args = [os.path.abspath(a) for a in sys.argv[1:] if not a.startswith('--')]
    for fname, rows in scan(args or None).items():
        flagged = [r for r in rows if show_all or r[3] <= 2 or r[2] > 0]
        print('%s: %d kernels, %d at <= 2 waves per SIMD or with scratch' % (fname, len(rows), len(flagged)))
        for (k, tot, scr, occ), pretty in zip(flagged, demangle([r[0] for r in flagged])):
            print('  waves %d  registers %3d  scratch %4d B  %s' % (occ, tot, scr, re.sub(r'\(anonymous namespace\)::', '', pretty)[:120]))
