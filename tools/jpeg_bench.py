"""bench.jpeg_leg on its own (f2 codec front-end: PIL vs host Huffman threads vs GPU entropy decode of restart intervals)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.jpeg_leg(), indent=1))
