#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Builds the reference's own CPU path of SuiteSparseQR
# (SPQR + the parts of CHOLMOD/AMD/COLAMD/CAMD/CCOLAMD/METIS it calls) from the sources
# where they lie under /root/reference, with plain gcc/g++ and no file of the reference
# copied, changed or stood in for.  Outputs go to oracle/_ref/ only.
#
# What makes it buildable in this image without any stand-in:
#   * the fork's SuiteSparseQR.hpp includes GPUQREngine.hpp -> <cuda_runtime.h> even in a
#     CPU build; the image holds NVIDIA's real header (shipped inside the triton wheel),
#     and it is only parsed for its types: nothing of CUDA is linked or run;
#   * BLAS/LAPACK = the image's MKL (/opt/conda/lib/libmkl_rt.so);
#   * CHOLMOD's supernodal *numeric* files (which reference openblas/cuda symbols in this
#     fork) are not on SPQR's call graph and are left out; every symbol resolves.
# Not the reference's Makefiles: the commands below are the whole recipe.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
OBJ=$OUT/obj
NVINC=$(python3 - <<'EOF'
import importlib.util, os
s = importlib.util.find_spec("triton")
print(os.path.join(os.path.dirname(s.origin), "backends", "nvidia", "include") if s else "")
EOF
)
[ -f "$NVINC/cuda_runtime.h" ] || { echo "no cuda_runtime.h in the image: reference unbuildable" >&2; exit 3; }
[ -d "$REF/SPQR/Source" ] || { echo "no reference tree at $REF" >&2; exit 3; }
mkdir -p $OBJ
J=${J:-8}
CF="-O2 -fPIC -w -DDLONG -DNDEBUG"
SSI="-I$REF/SuiteSparse_config"

list=$OBJ/cmds.txt; : > $list
add() { # add <obj-name> <compiler and flags...> <source>
  local o=$OBJ/$1.o; shift
  local src="${@: -1}"
  if [ ! -f $o ] || [ $src -nt $o ]; then echo "$* -o $o" >> $list; fi
}
add ssconfig gcc $CF $SSI -c $REF/SuiteSparse_config/SuiteSparse_config.c
for f in $REF/AMD/Source/amd_*.c; do add amd_$(basename $f .c) gcc $CF $SSI -I$REF/AMD/Include -c $f; done
for f in $REF/CAMD/Source/camd_*.c; do add camd_$(basename $f .c) gcc $CF $SSI -I$REF/CAMD/Include -c $f; done
add colamd gcc $CF $SSI -I$REF/COLAMD/Include -c $REF/COLAMD/Source/colamd.c
for f in $REF/CCOLAMD/Source/ccolamd.c; do add ccolamd gcc $CF $SSI -I$REF/CCOLAMD/Include -c $f; done
M=$REF/metis-5.1.0
MF="-O2 -fPIC -w -DLINUX -D_FILE_OFFSET_BITS=64 -DNDEBUG -DNDEBUG2 -DHAVE_EXECINFO_H -DHAVE_GETLINE -std=gnu99 -I$M/include -I$M/GKlib -I$M/libmetis"
for f in $M/GKlib/*.c; do add gk_$(basename $f .c) gcc $MF -c $f; done
for f in $M/libmetis/*.c; do add metis_$(basename $f .c) gcc $MF -c $f; done
CI="-I$REF/CHOLMOD/Include $SSI -I$REF/AMD/Include -I$REF/COLAMD/Include -I$REF/CAMD/Include -I$REF/CCOLAMD/Include -I$M/include"
for f in $REF/CHOLMOD/Core/cholmod_*.c $REF/CHOLMOD/Check/cholmod_*.c $REF/CHOLMOD/MatrixOps/cholmod_*.c \
         $REF/CHOLMOD/Partition/cholmod_*.c \
         $REF/CHOLMOD/Cholesky/cholmod_{amd,analyze,colamd,etree,postorder,rowcolcounts}.c \
         $REF/CHOLMOD/Supernodal/cholmod_super_symbolic.c; do
  add ch_$(basename $f .c) gcc $CF -DNMODIFY $CI -c $f
done
QI="$CI -I$REF/SPQR/Include -I$REF/GPUQREngine/Include -I$REF/SuiteSparse_GPURuntime/Include -I$NVINC"
for f in $REF/SPQR/Source/*.cpp; do add spqr_$(basename $f .cpp) g++ $CF -fopenmp $QI -c $f; done
add ref_driver g++ $CF -fopenmp $QI -c $HERE/ref_driver.cpp

if [ -s $list ]; then
  echo "compiling $(wc -l < $list) reference translation units"
  xargs -P $J -I{} sh -c "{}" < $list
fi
rm -f $OBJ/cmds.txt
g++ -shared -fopenmp -o $OUT/libspqr_ref.so $OBJ/*.o \
    -L/opt/conda/lib -lmkl_rt -Wl,-rpath,/opt/conda/lib -Wl,--no-undefined -lm -lrt -ldl -lpthread
echo "built $OUT/libspqr_ref.so"
