#!/bin/bash
# The mid-sized rounds of a folding step (2^19..2^22 pairs: tables that fit the 256 MB Infinity Cache) under the library's grid switches,
# one box, back to back: python tools/step_breakdown.py for cfg4's (ell = 26) and cfg3's (ell = 21) table.  usage: tools/sweep_sc_mid.sh OUT
out=${1:-/dev/stdout}; root=$(dirname $(dirname $(realpath $0)))
{
echo "# shipped"; python $root/tools/step_breakdown.py 26 | head -1; python $root/tools/step_breakdown.py 21 | head -1
for mp in 19 20 22 23; do echo "# REEF_SC_RANK1_MIN_POW=2^$mp"; REEF_SC_RANK1_MIN_POW=$((1 << mp)) python $root/tools/step_breakdown.py 26 | head -1; done
for it in 2 4 16 32; do echo "# REEF_SC_ITEMS=$it"; REEF_SC_ITEMS=$it python $root/tools/step_breakdown.py 26 | head -1; REEF_SC_ITEMS=$it python $root/tools/step_breakdown.py 21 | head -1; done
for fl in 128 512 1024; do echo "# REEF_SC_FLOOR=$fl"; REEF_SC_FLOOR=$fl python $root/tools/step_breakdown.py 26 | head -1; REEF_SC_FLOOR=$fl python $root/tools/step_breakdown.py 21 | head -1; done
for bl in 1024 4096 8192; do echo "# REEF_SC_BLOCKS=$bl"; REEF_SC_BLOCKS=$bl python $root/tools/step_breakdown.py 26 | head -1; REEF_SC_BLOCKS=$bl python $root/tools/step_breakdown.py 21 | head -1; done
} > $out 2>&1
