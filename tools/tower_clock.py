import sys, importlib, ctypes as C
sys.path.insert(0,'.')
P=importlib.import_module('alphazero-risk_amd')
L=P.load_library()
L.azr_debug_tower_clock.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_void_p,C.c_void_p]
for G in (256,512,2048):
    e=P.Engine(G,blocks=20,sims=100,dtype=P.NET_BF16); e.init_random(1); e.selfplay_start(1); e.selfplay_run(50)
    ghz=C.c_double(); ms=C.c_double()
    rc=L.azr_debug_tower_clock(e.h,G,3000 if G<2048 else 600,C.byref(ghz),C.byref(ms))
    print("G",G,"rc",rc,"sustained shader clock %.3f GHz"%ghz.value,"wg0 tower %.3f ms"%ms.value)
    e.close()
