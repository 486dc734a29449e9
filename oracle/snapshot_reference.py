"""TEST INFRASTRUCTURE -- the reference's Python package as ONE built artefact for the CPU-baseline leg on the GPU box.

    python oracle/snapshot_reference.py      /root/reference/stp3/**/*.py  ->  oracle/_ref/reference_stp3.zip (+ SNAPSHOT.json)

SURVEY.md section 8(d) asks for the REFERENCE's own modules, timed on the GPU node's host cores in the same job as the
GPU measurement.  /root/reference exists only in the build container, so ``__graft_entry__.build()`` -- which the driver
runs there every round -- packs the reference's ``stp3`` package (pure Python, no build step of its own) into a zip
archive under ``oracle/_ref/``: git-ignored (the reference's sources never enter this repository's history or its
working tree as files), not gpurun-ignored (the archive travels to the GPU box with the snapshot, like a built .so).
Python imports the package straight from the archive (zipimport).  Only ``bench.py``'s ``cpu_baseline`` leg and tests
import it, through ``oracle/ref_stubs.py``; nothing under ``st-p3_amd/`` does.  ``SNAPSHOT.json`` records the source path
and the sha256 of every member, so that a stale or edited archive is detectable (``verify()``)."""
import hashlib
import json
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE = '/root/reference'
DEST = os.path.join(HERE, '_ref')
ARCHIVE = 'reference_stp3.zip'


def snapshot(source=SOURCE, dest=DEST):
    """Pack ``source``/stp3/**/*.py (nothing else: no configs, images, maps) into ``dest``/reference_stp3.zip.  The
    reference's directories carry no ``__init__.py`` (namespace packages); the archive gets empty ones, which is what
    zipimport wants.  Returns the manifest."""
    pkg = os.path.join(source, 'stp3')
    if not os.path.isdir(pkg):
        raise RuntimeError(f'{pkg}: reference tree not present (the snapshot is taken in the build container only)')
    os.makedirs(dest, exist_ok=True)
    files, packages = {}, set()
    with zipfile.ZipFile(os.path.join(dest, ARCHIVE), 'w', zipfile.ZIP_DEFLATED) as z:
        for root, dirs, names in os.walk(pkg):
            dirs[:] = sorted(d for d in dirs if d != '__pycache__')
            for name in sorted(names):
                if not name.endswith('.py'):
                    continue
                src = os.path.join(root, name)
                rel = os.path.relpath(src, source)
                data = open(src, 'rb').read()
                z.writestr(zipfile.ZipInfo(rel, date_time=(2020, 1, 1, 0, 0, 0)), data)
                files[rel] = hashlib.sha256(data).hexdigest()
                packages.add(os.path.dirname(rel))
        for d in sorted(packages):
            init = os.path.join(d, '__init__.py')
            if init not in files:
                z.writestr(zipfile.ZipInfo(init, date_time=(2020, 1, 1, 0, 0, 0)), b'')
    manifest = {'source': source, 'archive': ARCHIVE, 'files': files,
                'note': 'verbatim members of the reference package for the cpu_baseline leg; git-ignored test infrastructure'}
    with open(os.path.join(dest, 'SNAPSHOT.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    return manifest


def archive_path(dest=DEST):
    return os.path.join(dest, ARCHIVE)


def verify(dest=DEST):
    """True when ``dest`` holds an archive whose members still carry the recorded hashes."""
    path = os.path.join(dest, 'SNAPSHOT.json')
    if not (os.path.isfile(path) and os.path.isfile(archive_path(dest))):
        return False
    manifest = json.load(open(path))
    try:
        with zipfile.ZipFile(archive_path(dest)) as z:
            return bool(manifest['files']) and all(hashlib.sha256(z.read(rel)).hexdigest() == sha
                                                   for rel, sha in manifest['files'].items())
    except (KeyError, zipfile.BadZipFile):
        return False


if __name__ == '__main__':
    m = snapshot()
    ok = verify()
    print(f'{len(m["files"])} files -> {archive_path()}', 'verified' if ok else 'VERIFY FAILED')
    sys.exit(0 if ok else 1)
