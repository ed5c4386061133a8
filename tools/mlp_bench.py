import sys, json
sys.path.insert(0, "/root/repo")
import bench
print(json.dumps(bench.mlp_rate(), indent=1))
