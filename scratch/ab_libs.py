import sys, os, subprocess
libs = sorted(f for f in os.listdir('scratch') if f.startswith('libprt_') and f.endswith('.so'))
code = '''
import sys, torch
sys.path.insert(0,'.')
from pyrate_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from pyrate_amd import engine, systems
dev=torch.device('cuda',0)
sysd=engine.DeviceSystem(systems.double_gauss_records(),0)
x0,k0,e0,_=systems.double_gauss_bundle_device(10**7,dev)
bufs=sysd.alloc_outputs(x0.shape[1],0)
for _ in range(30): sysd.trace_into(x0,k0,bufs,e0)
torch.cuda.synchronize()
print(sys.argv[1].split('_')[-1], ["%.4f"%sysd.trace_timed(x0,k0,bufs,50,e0) for _ in range(3)])
'''
for rep in range(2):
    for l in libs:
        subprocess.run([sys.executable, '-c', code, os.path.join('scratch', l)])
