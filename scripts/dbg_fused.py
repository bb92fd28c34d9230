import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import weights as W
from oracle import nerf_oracle as O
from nerf_amd import ops
from nerf_amd.mip_model import MipNeRF
mip=MipNeRF(10,4,256); mip.load_state_dict(W.mip_state("he")); mip=mip.cuda().eval()
n,S=8,128
g=torch.Generator().manual_seed(1)
pose=O.pose_spherical(10.0,-30.0,4.0)[:3]
dirs=O.ray_dirs_image(pose,40,40,O.fov2focal(0.6911112070083618,(40,40))).reshape(-1,3)
rays=torch.cat((pose[:,-1].expand(n,-1),dirs[torch.randint(0,1600,(n,),generator=g)]),-1).contiguous().cuda()
z=torch.sort(2+4*torch.rand(n,S+1,generator=g),dim=-1)[0].cuda()
for P in (ops.BF16,):
    rgb,depth,w=ops.mip_forward_composite(mip.packed(P),P,rays,z,S,True,2.0,6.0,want_depth=True,want_weights=True)
    rgbo=ops.mip_forward_samples(mip.packed(P),P,ops.samples_rays(rays,S,z=z),(n,S),"cuda")
    rgb2,w2,depth2,_=ops.composite(rgbo,z,rays,True,True,ops.ACT_RELU,(2.0,6.0))
    torch.set_printoptions(precision=4,linewidth=200)
    print("rgb fused", rgb[:4]); print("rgb ref", rgb2[:4])
    e=(w-w2).abs()
    print("w err per 32-block of ray0:", [float(e[0,k*32:(k+1)*32].max()) for k in range(4)])
    print("w fused ray0 [0:4],[32:36],[64:68],[96:100]", w[0,0:4], w[0,32:36], w[0,64:68], w[0,96:100])
    print("w ref   ray0", w2[0,0:4], w2[0,32:36], w2[0,64:68], w2[0,96:100])
    print("ratio seg1", (w[0,32:40]/w2[0,32:40]), "seg2", (w[0,64:72]/w2[0,64:72]), "seg3", (w[0,96:104]/w2[0,96:104]))
