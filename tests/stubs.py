"""Test doubles (not the oracle): a deterministic stand-in for the pipeline protocol so that sampler /
runner logic can be exercised on CPU and across processes."""
import torch

from diffuman4d_amd.host.schedule import plan_sweep


class StubPipeline:
    """Implements the pipeline protocol.  'Denoising' adds 1.0 to a target latent per
    step, so the final value of a cell counts how many steps it received; cond rows carry -1."""

    def __init__(self, h=2, w=2):
        self.device = torch.device("cpu")
        self.h, self.w = h, w
        self.calls = []

    def sliding_iterative_denoise(self, pixel_values, plucker_embeds, skeletons, cond_masks, latents, domain,
                                  timestep_indices, window_size, sliding_stride, sliding_shift, bidirectional,
                                  num_denoising_steps, alternation_rounds, guidance_scale, tqdm=None):
        n = pixel_values.shape[0]
        cond = (cond_masks[:, 0, 0, 0] == 0).tolist()
        plan = plan_sweep(cond, timestep_indices.tolist(), domain, window_size, sliding_stride, sliding_shift,
                          bidirectional, num_denoising_steps, alternation_rounds)
        self.calls.append({"latents_was_none": latents is None, "domain": domain, "n": n,
                           "cond_rows": [i for i, c in enumerate(cond) if c]})
        lat = torch.zeros(n, 4, self.h, self.w) if latents is None else latents.clone().float()
        for w, c in zip(plan.windows, plan.is_cond):
            lat[w[c]] = -1.0
            lat[w[~c]] += 1.0
        tidx = torch.from_numpy(plan.final_timestep_indices)
        return {"images": torch.zeros(n, 3, 8 * self.h, 8 * self.w), "latents": lat, "timestep_indices": tidx,
                "fully_denoised": tidx == plan.num_inference_steps}


class StackStubPipeline(StubPipeline):
    """StubPipeline that also takes the runner's task stacks (runner.task_batch): every task of a stack gets what it gets alone, and the
    stack sizes are recorded."""

    def __init__(self, h=2, w=2):
        super().__init__(h, w)
        self.stacks = []

    def sliding_iterative_denoise_stack(self, tasks, domain, tqdm=None, decode="all", **sweep):
        self.stacks.append(len(tasks))
        return [self.sliding_iterative_denoise(domain=domain, tqdm=tqdm, **t, **sweep) for t in tasks]


class ShardStubPipeline(StubPipeline):
    """StubPipeline that also takes the runner's frame-shard extensions: with `shard` (parallel.FrameShard) every rank of the
    group updates ITS frames of each window and the rows are all-gathered (a real collective of the test's backend), as
    Diffuman4DPipeline.denoise_latents does; the result must equal the unsharded stub's on every rank of the group."""

    def sliding_iterative_denoise(self, pixel_values, plucker_embeds, skeletons, cond_masks, latents, domain,
                                  timestep_indices, window_size, sliding_stride, sliding_shift, bidirectional,
                                  num_denoising_steps, alternation_rounds, guidance_scale, tqdm=None, shard=None, noise_seed=None,
                                  decode="all"):
        if shard is not None:  # the non-leading ranks of a shard group decode nothing (runner._run_sharded), the leader everything
            assert decode == ("all" if shard.rank == 0 else "none"), (shard.rank, decode)
        if shard is None:
            return super().sliding_iterative_denoise(pixel_values, plucker_embeds, skeletons, cond_masks, latents, domain,
                                                     timestep_indices, window_size, sliding_stride, sliding_shift, bidirectional,
                                                     num_denoising_steps, alternation_rounds, guidance_scale, tqdm)
        assert noise_seed is not None, "a shard group must be handed a common noise seed"
        n = pixel_values.shape[0]
        cond = (cond_masks[:, 0, 0, 0] == 0).tolist()
        plan = plan_sweep(cond, timestep_indices.tolist(), domain, window_size, sliding_stride, sliding_shift,
                          bidirectional, num_denoising_steps, alternation_rounds)
        self.calls.append({"latents_was_none": latents is None, "domain": domain, "n": n, "sharded": shard.world, "noise_seed": noise_seed,
                           "cond_rows": [i for i, c in enumerate(cond) if c]})
        lat = torch.zeros(n, 4, self.h, self.w) if latents is None else latents.clone().float()
        for w, c in zip(plan.windows, plan.is_cond):
            sl = shard.local_frames(len(w))
            wl, cl = torch.as_tensor(w[sl]), torch.as_tensor(c[sl])
            new = lat[wl].clone()
            new[cl] = -1.0
            new[~cl] += 1.0
            lat[torch.as_tensor(w)] = shard.gather_rows(new)
        tidx = torch.from_numpy(plan.final_timestep_indices)
        return {"images": torch.zeros(n, 3, 8 * self.h, 8 * self.w), "latents": lat, "timestep_indices": tidx,
                "fully_denoised": tidx == plan.num_inference_steps}
