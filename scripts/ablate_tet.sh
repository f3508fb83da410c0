#!/bin/bash
# stage ablation of the resident dense kernel (PA_DBG bits: 1 no x gather, 2 q-data from one block,
# 4 no MFMA work, 8 E-vector store to one slot, 16 no curl-orientation exchange)
export PALACE_AMD_LIB=$(pwd)/palace_amd/lib/libpalace_amd_ablate.so
for dbg in 0 1 2 4 8 16 3 7 15 31; do
  echo -n "PA_DBG=$dbg  "; PA_DBG=$dbg N=${N:-36} python scripts/time_tet.py 2>&1 | grep "^curl " 
done
