"""Body of tests/test_gpu_parity.py::test_data_parallel_minibatch_step_is_captured_with_its_all_reduce, run as its own process."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    import torch.distributed as dist
    from steppingstone_amd import ppo
    dev = torch.device("cuda:0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29650 + os.getpid() % 300), rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(3)
        data = (torch.randn(4096, 60, device=dev), torch.randn(4096, 21, device=dev).clamp(-1, 1), torch.randn(4096, 1, device=dev),
                torch.randn(4096, 1, device=dev), -20 + torch.randn(4096, 1, device=dev), torch.randn(4096, 1, device=dev))
        finals = []
        for use_graph in (False, True):
            torch.manual_seed(5)
            ac = ppo.ActorCritic(num_ensembles=1).to(dev)
            agent = ppo.PPO(ac, mini_batch_size=512, use_graph=use_graph, graph_collectives=True, force_collective=True)
            g = torch.Generator(device=dev)
            g.manual_seed(11)
            for k in range(8):
                idx = torch.randperm(4096, device=dev, generator=g)[:512]
                out = agent._graph_step(data, idx, refresh=(k == 0)) if use_graph else torch.stack(agent._gathered_step(data, idx))
            if use_graph:
                assert agent._graph_ok() and agent._graph is not None and agent.graph_fallback is None, agent.graph_fallback
            finals.append((torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).clone(), out.clone()))
        assert torch.allclose(finals[0][0], finals[1][0], atol=5e-4), float((finals[0][0] - finals[1][0]).abs().max())
        assert torch.allclose(finals[0][1], finals[1][1], rtol=1e-2, atol=1e-4)
        torch.cuda.synchronize()
        print("dp graph capture ok")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
