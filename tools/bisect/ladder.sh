#!/bin/bash
# ladder.sh LO HI STEPS: builds sc_bNNNNN.so from dd2.hip (16-byte slot reads) for STEPS limits of -opt-bisect-limit between LO
# and HI, in parallel; `python run2.py` on the GPU box then shows where the first bad build is.
cd "$(dirname "$0")"
F="-std=c++17 -fPIC --offload-arch=gfx950 -I../../reef_amd/csrc -Wno-unused-function -shared -w -O3 -DREEF_VEC_VARIANT=2 -DREEF_COOP_WIDE_READS -fno-slp-vectorize -mllvm -amdgpu-load-store-vectorizer=0"
rm -f sc_b*.so
for k in $(seq 0 $3); do
  L=$(( $1 + ($2 - $1) * k / $3 ))
  ( hipcc $F -mllvm -opt-bisect-limit=$L dd2.hip -o $(printf "sc_b%05d.so" $L) 2>/dev/null ) &
  if (( (k+1) % 8 == 0 )); then wait; fi
done
wait
ls sc_b*.so | wc -l
