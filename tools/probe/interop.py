import ctypes, os, torch
here=os.path.dirname(os.path.abspath(__file__))
lib=ctypes.CDLL(os.path.join(here,'libinterop.so'))
lib.probe_axpy.argtypes=[ctypes.c_void_p,ctypes.c_void_p,ctypes.c_float,ctypes.c_int,ctypes.c_void_p]
x=torch.arange(1000,device='cuda',dtype=torch.float32); y=torch.ones(1000,device='cuda')
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    rc=lib.probe_axpy(x.data_ptr(),y.data_ptr(),2.0,1000,torch.cuda.current_stream().cuda_stream)
s.synchronize()
print('rc',rc,'ok',bool(torch.allclose(y,1+2*x)))
import subprocess
print(subprocess.run(['bash','-c',f'grep -E "amdhip|hsa-runtime" /proc/{os.getpid()}/maps | awk "{{print \\$6}}" | sort -u'],capture_output=True,text=True).stdout)
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).total_memory/1e9)
