import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s}  total {float(r['TotalDurationNs']) / 1e6:9.3f} ms  avg {float(r['AverageNs']) / 1e3:9.1f} us")
