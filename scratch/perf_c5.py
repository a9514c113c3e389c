import sys; sys.path.insert(0, '.')
import torch, bench
print(bench.grouped_c5(torch.device("cuda:0"), 20))
