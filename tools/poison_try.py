import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy, torch, poison
torch.zeros(1, device='cuda')
p0 = poison.probe()
print('before: share of words equal to the pattern: v %.3f a %.3f lds %.3f' % tuple((p0[:, i] == 0x7fc0babe).mean() for i in range(3)))
poison.poison(); torch.cuda.synchronize()
p1 = poison.probe()
print('after : share of words equal to the pattern: v %.3f a %.3f lds %.3f' % tuple((p1[:, i] == 0x7fc0babe).mean() for i in range(3)))
