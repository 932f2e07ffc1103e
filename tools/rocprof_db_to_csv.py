#!/usr/bin/env python3
"""Dump the per-kernel summary (`top_kernels` view) of a rocprofv3 results .db to CSV."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for r in cur:
    w.writerow(r)
