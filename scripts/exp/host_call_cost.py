"""host-side cost of the calls every op launch makes (the GPU box's python): torch.cuda.current_stream() against the raw-stream getter"""
import os, timeit, torch
torch.cuda.init()
x = torch.zeros(1, device="cuda")
n = 20000
print("environment variables:", len(os.environ))
print("os.environ.get                          %.2f us" % (timeit.timeit(lambda: os.environ.get("PYTORCH_NVML_BASED_CUDA_CHECK"), number=n) / n * 1e6))
print("torch.cuda.is_available                 %.2f us" % (timeit.timeit(torch.cuda.is_available, number=n) / n * 1e6))
print("torch.cuda.current_stream().cuda_stream %.2f us" % (timeit.timeit(lambda: torch.cuda.current_stream().cuda_stream, number=n) / n * 1e6))
print("torch._C._cuda_getCurrentRawStream(0)   %.2f us" % (timeit.timeit(lambda: torch._C._cuda_getCurrentRawStream(0), number=n) / n * 1e6))
print("torch.cuda.current_device               %.2f us" % (timeit.timeit(torch.cuda.current_device, number=n) / n * 1e6))
print("torch.empty(16, device=cuda)            %.2f us" % (timeit.timeit(lambda: torch.empty(16, device="cuda"), number=n) / n * 1e6))
print("x.data_ptr()                            %.2f us" % (timeit.timeit(x.data_ptr, number=n) / n * 1e6))
