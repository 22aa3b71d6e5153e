"""Pretty-prints a UFD_BENCH_DUMP json (per-launch-site device times from ufd_profile_read)."""
import json, sys
d = json.load(open(sys.argv[1]))
steps = d['steps']
rows = sorted(d['stats'], key=lambda s: -s['total_ms'])
tot = sum(s['total_ms'] for s in rows)
print('total gpu ms/step %.3f' % (tot / steps))
for s in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    ms = s['total_ms'] / max(s['launches'], 1)
    print('%-36s %8.1f us  %7.1f GB/s %7.2f TF/s  %5.1f%%' % (s['name'], ms * 1e3, s['bytes'] / max(s['launches'], 1) / ms / 1e6 if ms else 0,
          s['flops'] / max(s['launches'], 1) / ms / 1e9 if ms else 0, 100 * s['total_ms'] / tot))
