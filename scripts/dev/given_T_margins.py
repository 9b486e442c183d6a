import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from tests.test_gs2d_gpu import given_T_check, FUZZ_2D
from tests.test_oracle2d_cpu import make_case2d
dev=torch.device('cuda:0')
for k in list(range(8))+[26]:
    rep={}
    case=FUZZ_2D[k]
    given_T_check(make_case2d(**case)[0], case["seed"], dev, report=rep)
    print(k, {n:(f"{v[0]:.1e}",f"{v[1]:.1e}") for n,v in rep.items()})
