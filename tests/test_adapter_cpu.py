"""The three translation units of adapter/ are what an SPQR build compiles in place of SPQR/SPQRGPU.
They include SPQR's own headers, which this repository does not hold; where the reference tree is
present (not on the GPU box) they are compiled against those headers where they lie, syntax only:
the signatures must be the ones SPQR/Include/spqrgpu.hpp:6-82 declares, this fork's extra QRsym
argument included, and include/GPUQREngine.hpp must stand in for the reference's header of that name
inside SuiteSparseQR.hpp (Front and Workspace pointers in spqr_symbolic, the Int macro of
cholmod_subtree.h)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
UNITS = sorted(glob.glob(os.path.join(ROOT, "adapter", "*.cpp")))


def test_there_are_three_units():
    assert [os.path.basename(u) for u in UNITS] == [
        "spqrgpu_buildAssemblyMaps.cpp", "spqrgpu_computeFrontStaging.cpp", "spqrgpu_kernel.cpp"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "SPQR", "Include")) or shutil.which("g++") is None,
                    reason="needs the reference tree's headers and g++")
@pytest.mark.parametrize("unit", UNITS, ids=[os.path.basename(u) for u in UNITS])
def test_adapter_unit_compiles_against_spqr_headers(unit):
    cmd = ["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-DSUITESPARSE_HIP",
           "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(REF, "SPQR", "Include"),
           "-I", os.path.join(REF, "CHOLMOD", "Include"),
           "-I", os.path.join(REF, "SuiteSparse_config"), unit]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    # without the gate the units are empty: an SPQR build that does not ask for them is untouched
    r = subprocess.run([c for c in cmd if c != "-DSUITESPARSE_HIP"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


CALLER = r"""
#include "spqr.hpp"
#include "spqrgpu.hpp"          // the reference's declarations (it pulls "GPUQREngine.hpp": this repository's)
void call_them (spqr_blob <double> *Bd, spqr_blob <Complex> *Bc, cholmod_common *cc)
{
    spqrgpu_kernel (Bd) ;
    spqrgpu_kernel (Bc) ;
    spqrgpu_computeFrontStaging (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
#ifdef EXCLUDE_ALLOCATION_TIME
        (spqr_symbolic *) 0,
#endif
        cc) ;
    spqrgpu_buildAssemblyMaps (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) ;
}
#ifdef EXCLUDE_ALLOCATION_TIME
// what SPQR/Source/spqr_1factor.cpp:643-655 does with the workspaces of the symbolic object once the
// numeric factorization is done, statement for statement: it must compile against this repository's
// GPUQREngine.hpp, because that is where the AMD driver's symbolic object is released
void release_them (spqr_symbolic *QRsym, cholmod_common *cc)
{
    QRsym->wsMondoS = Workspace::destroy(QRsym->wsMondoS);
    QRsym->wsRjmap  = Workspace::destroy(QRsym->wsRjmap);
    QRsym->wsRimap  = Workspace::destroy(QRsym->wsRimap);
    QRsym->fronts[0] = (Front*) cholmod_l_free (QRsym->maxfronts_in_stage, sizeof(Front), QRsym->fronts[0], cc) ;
    if (QRsym->fronts[1] != NULL) QRsym->fronts[1] = (Front*) cholmod_l_free (QRsym->maxfronts_in_stage, sizeof(Front), QRsym->fronts[1], cc) ;
    QRsym->wsMondoF = Workspace::destroy(QRsym->wsMondoF);
    QRsym->wsMondoR[0] = Workspace::destroy(QRsym->wsMondoR[0]);
    if (QRsym->wsMondoR[1] != NULL) QRsym->wsMondoR[1] = Workspace::destroy(QRsym->wsMondoR[1]);
    QRsym->wsS = Workspace::destroy(QRsym->wsS);
}
#endif
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "SPQR", "Include")) or shutil.which("g++") is None
                    or shutil.which("nm") is None, reason="needs the reference tree's headers, g++ and nm")
def test_adapter_units_define_what_spqr_headers_declare(tmp_path):
    """A caller compiled against the reference's own declarations (SPQR/Include/spqrgpu.hpp) must find
    its four symbols, by their mangled names, among those the adapter units define: a parameter list
    that strays from the header's would define another overload and leave the caller's undefined."""
    inc = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(REF, "SPQR", "Include"),
           "-I", os.path.join(REF, "CHOLMOD", "Include"), "-I", os.path.join(REF, "SuiteSparse_config")]
    src = tmp_path / "caller.cpp"
    src.write_text(CALLER)
    subprocess.run(["g++", "-std=c++11", "-c", str(src), "-o", str(tmp_path / "caller.o")] + inc,
                   check=True, capture_output=True, timeout=300)
    wanted = {l.split()[-1] for l in subprocess.run(["nm", "-u", str(tmp_path / "caller.o")], check=True,
              capture_output=True, text=True).stdout.splitlines() if "spqrgpu_" in l}
    assert len(wanted) == 4, wanted
    defined = set()
    for u in UNITS:
        obj = tmp_path / (os.path.basename(u) + ".o")
        subprocess.run(["g++", "-std=c++11", "-DSUITESPARSE_HIP", "-c", u, "-o", str(obj)] + inc,
                       check=True, capture_output=True, timeout=300)
        defined |= {l.split()[-1] for l in subprocess.run(["nm", "--defined-only", str(obj)], check=True,
                    capture_output=True, text=True).stdout.splitlines()}
    assert wanted <= defined, wanted - defined
