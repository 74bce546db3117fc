import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import faulthandler; faulthandler.enable()
import test_gpu_graph_replay as T
getattr(T, sys.argv[1])(torch.device("cuda:0"))
print("OK", sys.argv[1])
