import sys, json
sys.path.insert(0, '/root/repo')
import bench
from glomap_amd import _lib
ctx = _lib.Context(0)
print(json.dumps(bench.bench_filters(ctx), indent=1))
