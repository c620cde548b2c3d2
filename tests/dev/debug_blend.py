import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gance_amd import hip_lib, synthetic
from oracle import audio_ref
g = np.load('tests/golden/blend_n60_seed0_roll_k3.npz')
audio = synthetic.synthetic_audio(60, 512, seed=0)
lat = synthetic.synthetic_final_latents(30, 512, seed=4)
b = hip_lib.Blend(60, 30, 0.25, True, (-5,5), 12, 3)
da = torch.from_numpy(audio).cuda(); dr = torch.from_numpy(np.ascontiguousarray(lat[0])).cuda()
b.run_device(da.data_ptr(), audio.size, dr.data_ptr(), 0, 0, True, 0); torch.cuda.synchronize()
r = b.read_stage('raw_rms'); w = g['raw_rms']
bad = np.nonzero(r != w)[0]; print('bad', bad, r[bad], w[bad], r[bad].view(np.int32) - w[bad].view(np.int32))
sq = (audio.reshape(60,512)**2)
s = np.array([audio_ref.numpy_pairwise_sum_f32(sq[t]) for t in range(60)])
print('mean/sqrt check', np.array_equal(np.sqrt(s/np.float32(512)), w))
for name in ['rolling_average','rolling_smoothed']:
    d = b.read_stage(name) - g[name]; print(name, np.abs(d).max())
print('roll', np.array_equal(b.read_stage('roll_values'), g['roll_values']), np.array_equal(b.read_stage('network_indices'), g['network_indices']))
for st,key in [('scaled','scaled'),('smoothed_time','smoothed_time'),('smoothed','smoothed'),('final','final'),('blend_row','combined_row0')]:
    got = b.read_stage(st).reshape(-1)[::13]; print(st, np.abs(got - g[key+'_sample']).max())
db = b.read_stage('db').T.reshape(-1)[::13]; print('db', np.abs(db - g['db_sample']).max())
