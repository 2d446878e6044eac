#!/bin/bash
# A/B build: tools/ab/libpyscf_amd_ab.so = the product library with df_jk.hip compiled -DPAMD_AB_VARIANT (guard the variant under test with it).

set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" 2>/dev/null
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPAMD_AB_VARIANT -c pyscf_amd/csrc/df_jk.hip -o tools/ab/df_jk_ab.o
objs=$(ls pyscf_amd/build/*.o | grep -v "/df_jk.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libpyscf_amd_ab.so $objs tools/ab/df_jk_ab.o
ls -la tools/ab/libpyscf_amd_ab.so
