import csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step", round(tot / steps / 1e6, 2))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    print("%-72s calls/step=%6.1f ms/step=%6.2f avg_us=%7.1f" % (r["Name"][:72], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6, float(r["AverageNs"]) / 1e3))
