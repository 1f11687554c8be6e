"""usage (GPU box): python profiles/microbench/td3_soak.py — 1200 graph-replayed closed-loop TD3 steps of 4096 environments (BASELINE config 5), a line per 100 steps: what the step costs once episodes end by reaching the goal (the bench window sits in the first ~140 steps), and that nothing non-finite appears.  TD3_PIPELINE=1: the opt-in pipelined loop (BatchedTD3(pipeline=True))."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc4rl_amd import BatchedCartPoleSwingUpEnv, BatchedTD3, cartpole_ocp
E = 4096
env = BatchedCartPoleSwingUpEnv(E, device="cuda", seed=0)
ag = BatchedTD3(cartpole_ocp(), env, batch_size=E, buffer_steps=64, policy_delay=2, lr_actor=1e-6, seed=0, replay_iterates=True, pipeline=bool(int(os.environ.get('TD3_PIPELINE', '0'))))
ag.collect(4)
ag.enable_graphs()
th0 = ag.theta.clone()
for blk in range(12):
    ag.collect(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ag.step(100)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    st = ag.last_stats()
    crit = torch.cat([p.detach().reshape(-1) for p in ag.critic.parameters()])
    print("steps %4d: %.3f ms/step, mean reward %.4f, converged %.4f, episodes ended %d, critic loss %.4f, critic finite %s, |theta - theta0| %.2e, env steps max %d" % (
        (blk + 1) * 100, dt * 1e3, st["mean_reward"], st["converged_fraction"], st["episodes_ended"], ag.last_critic_loss(), bool(torch.isfinite(crit).all()),
        float((ag.theta - th0).abs().max()), int(env.steps.max())), flush=True)
assert bool(torch.isfinite(ag.theta).all()) and bool(torch.isfinite(ag.buffer.data).all())
print("soak ok")
