import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np
import gnuspeech_amd as g, cases, oracle_lib as O, golden_io
np.set_printoptions(linewidth=200, precision=6)
for name in (sys.argv[1:] or ['monet_vowel_44k','tract_vowel_1s','gnuspeech_window_44k']):
    gold=golden_io.load(name)
    pd=gold['params_dict']
    b=g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    fr=gold['frames']
    pcm,ns,mx=b.synthesize([fr])
    o=O.synthesize(gold['params'], fr.astype(np.float32).astype(np.float64))
    x=pcm[0].astype(np.float64); r=o['samples']; m=o['maximumSampleValue']
    print(name,'N',ns[0],o['numberSamples'],'max',mx[0],m)
    e=(x-r)/m
    print(' rms',np.sqrt(np.mean(e**2)),'maxabs',np.abs(e).max(),'argmax',np.abs(e).argmax())
    seg=len(e)//10
    print(' rms per tenth:',[float('%.2e'%np.sqrt(np.mean(e[i*seg:(i+1)*seg]**2))) for i in range(10)])
    print(' gpu[:12]',x[:12]/m); print(' ref[:12]',r[:12]/m)
    k=np.abs(e).argmax(); print(' around max: gpu',x[k-3:k+4]/m,'ref',r[k-3:k+4]/m)
    nz=np.nonzero(np.abs(e)>1e-4)[0]; print(' first idx with err>1e-4:', nz[:5], 'count',len(nz))
b=g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params()))
nt=b.noise_table(64); ref=O.lp_noise(64).astype(np.float32)
print('noise gpu',nt[:8]); print('noise ref',ref[:8]); print('noise equal:',np.array_equal(nt,ref), np.nonzero(nt!=ref)[0][:5])
