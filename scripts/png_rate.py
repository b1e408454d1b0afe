"""The PNG input path's decode rates on this host (bench.py's `boundary.png_decode` leg alone; no GPU needed)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.png_decode_rate(n_images=int(sys.argv[1]) if len(sys.argv) > 1 else 64), indent=1))
