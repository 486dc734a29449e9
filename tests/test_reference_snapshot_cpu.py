"""CPU: the reference package as a built artefact (oracle/snapshot_reference.py) -- what bench.py's cpu_baseline leg
imports on the GPU box, where /root/reference does not exist.  Container only (needs the mounted reference to pack)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import snapshot_reference  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(snapshot_reference.SOURCE, 'stp3')),
                                reason='the reference tree is mounted in the build container only')

PROBE = r'''
import json, sys, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/st-p3_amd')
from oracle import ref_stubs
ref_stubs.SNAPSHOT_ROOT = %(dest)r
ref_stubs.install()
import stp3.utils.geometry as g
v = torch.linspace(-0.3, 0.4, 12).view(2, 6)
print(json.dumps({'file': g.__file__, 'root': ref_stubs.reference_root(), 'mat': g.pose_vec2mat(v).flatten().tolist()}))
'''


def _probe(dest, reference_root):
    env = dict(os.environ, STP3_REFERENCE_ROOT=reference_root)
    out = subprocess.run([sys.executable, '-c', PROBE % {'root': ROOT, 'dest': dest}], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-800:]
    return json.loads(out.stdout.splitlines()[-1])


def test_archive_verifies_and_imports_like_the_mounted_reference(tmp_path):
    dest = str(tmp_path / '_ref')
    manifest = snapshot_reference.snapshot(dest=dest)
    assert snapshot_reference.verify(dest) and 'stp3/models/stp3.py' in manifest['files']
    assert sorted(os.listdir(dest)) == ['SNAPSHOT.json', snapshot_reference.ARCHIVE]          # one archive, no source files
    from_archive = _probe(dest, '/nonexistent')
    mounted = _probe(dest, snapshot_reference.SOURCE)
    assert from_archive['root'].endswith(snapshot_reference.ARCHIVE) and snapshot_reference.ARCHIVE in from_archive['file']
    assert mounted['root'] == snapshot_reference.SOURCE
    assert from_archive['mat'] == mounted['mat']                                               # same code, same bits


def test_a_tampered_archive_is_refused(tmp_path):
    dest = str(tmp_path / '_ref')
    snapshot_reference.snapshot(dest=dest)
    man = json.load(open(os.path.join(dest, 'SNAPSHOT.json')))
    first = sorted(man['files'])[0]
    man['files'][first] = '0' * 64
    json.dump(man, open(os.path.join(dest, 'SNAPSHOT.json'), 'w'))
    assert not snapshot_reference.verify(dest)
