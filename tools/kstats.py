"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory: name, calls, average ns."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(6), '%9.1f us' % (float(r['AverageNs']) / 1e3), r['Percentage'])
