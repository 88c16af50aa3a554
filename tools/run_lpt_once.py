import sys, torch
sys.path.insert(0, "/root/repo")
from closed_chain_motion_planner_amd import Context, KinematicChainConstraint
ctx = Context(0)
c = KinematicChainConstraint.from_yaml("/root/repo/tests/golden/config/Wine_Bottle.yaml", ctx=ctx)
q = c.ambient_uniform_batch(0xC3, 0, 262144)
ctx.set_lpt(2, 0)
for _ in range(4):
    c.project_batch(q)
torch.cuda.synchronize()
