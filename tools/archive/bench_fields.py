"""Reads bench.py JSON lines on stdin, prints the fields that matter when sweeping options."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    c = d["config"]
    print(f"streams {c['streams']} c {c['window_bits']} W {c['windows']} G {c['bucket_groups']} ms/step {d['ms_per_step']:.3f} "
          f"Mpairs/s {d['value'] / 1e6:.1f} accum_ms {d['roofline']['kernel_ms']:.3f} {' '.join(sys.argv[1:])}")
