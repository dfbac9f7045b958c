import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
import stark_rs_amd as s
def mine():
    e = s.Engine(); print("engine ok", e.prim_nth_root(8)); return e
def theirs():
    import torch
    torch.cuda.init(); x = torch.ones(4, device="cuda"); print("torch ok", float(x.sum()))
if order == "mine_first":
    e = mine(); theirs()
else:
    theirs(); e = mine()
print("both ok", order)
