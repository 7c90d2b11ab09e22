"""HIP-graph replay of a fixed launch sequence (the launch-bound legs: one image of the evaluation loop is ~60 kernel launches
of 5-400 us each, the decoder's small ones back to back).

    g = GraphedCall(fn, *example_tensors)      # runs fn twice to warm up, then captures it once (torch.cuda.CUDAGraph)
    out = g(*new_tensors)                      # copies the new values into the captured inputs, replays, returns fn's outputs

`fn` must be a pure function of its tensor arguments with fixed shapes: every e3dge_* launch goes to torch's current stream,
allocates through torch's allocator only and never synchronises, so the whole renderer / texture head / decoder forward is
capturable (no host-side reads: the modulated convolutions take their operand scale from device-side amax buffers).  The
outputs are the captured tensors themselves -- valid until the next call; clone what must survive."""
import torch


class GraphedCall:
    def __init__(self, fn, *example_inputs, warmup=2):
        if not example_inputs or not all(torch.is_tensor(t) and t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedCall needs GPU tensors as example inputs")
        self._static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                      # caches (weight images, FIR kernels, cuBLAS handles) are built outside the capture
                fn(*self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # captured on the stream the warm-up ran on: per-stream scratch (the decoder's modulation tables) already exists there
        with torch.cuda.graph(self.graph, stream=side), torch.no_grad():
            self.outputs = fn(*self._static_in)

    def __call__(self, *inputs):
        if len(inputs) != len(self._static_in):
            raise RuntimeError(f"GraphedCall captured {len(self._static_in)} inputs, got {len(inputs)}")
        for dst, src in zip(self._static_in, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise RuntimeError(f"GraphedCall: input {tuple(src.shape)} {src.dtype} does not match the captured "
                                   f"{tuple(dst.shape)} {dst.dtype}")
            dst.copy_(src)
        self.graph.replay()
        return self.outputs
