"""capture one cal_loss + backward of a model into a hipGraph outside the Trainer (a user doing it by hand); run with python -X faulthandler"""
import os, sys, torch, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from sslrec_amd.config.configurator import configs, load_config
from sslrec_amd.data_utils.data_handler_general_cf import DataHandlerGeneralCF
from sslrec_amd.data_utils.synth import make_dataset
from sslrec_amd.models.bulid_model import build_model
model_name, graph_name, d = sys.argv[1], sys.argv[2], int(sys.argv[3])
dev = 'cuda:0'
raw = make_dataset(graph_name)
trn = sp.coo_matrix((raw != 0).astype(np.float32))
load_config(model_name, device=dev, overrides={'data': {'synthetic': 'tiny'}, 'model': {'embedding_size': d, 'device_rng': True, 'keep_rate': float(sys.argv[4])}})
dh = DataHandlerGeneralCF(); dh.trn_mat = trn
configs['data']['user_num'], configs['data']['item_num'] = trn.shape
dh.torch_adj = dh._make_torch_adj(trn).to(dev)
torch.manual_seed(0)
model = build_model(dh).to(dev)
B = 4096
gen = torch.Generator().manual_seed(1)
batch = [torch.randint(0, trn.shape[0], (B,), generator=gen).to(dev), torch.randint(0, trn.shape[1], (B,), generator=gen).to(dev), torch.randint(0, trn.shape[1], (B,), generator=gen).to(dev)]
def step():
    model.zero_grad(set_to_none=True)
    loss, _ = model.cal_loss(batch)
    loss.backward()
    return loss
if os.environ.get('PROBE_NO_DEFAULT_STREAM') != '1':
    for _ in range(3): step()
    torch.cuda.synchronize(); print('eager ok', flush=True)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize(); print('side-stream warm-up ok', flush=True)
g = torch.cuda.CUDAGraph()
model.zero_grad(set_to_none=True)
mode = sys.argv[5] if len(sys.argv) > 5 else ''
opt = torch.optim.Adam(model.parameters(), lr=1e-3) if mode == 'adam' else None
kw = {'capture_error_mode': 'thread_local'} if mode == 'tl' else {}
if mode == 'train':
    model.train()
with torch.cuda.graph(g, **kw):
    loss_g, _ = model.cal_loss(batch)
    print('captured cal_loss', flush=True)
    loss_g.backward()
    print('captured backward', flush=True)
    if opt is not None:
        opt.step()
    if mode == 'join':      # end the captured region with work on the capture stream that reads every gradient
        tot = sum(p.grad.sum() for p in model.parameters())
print('capture closed', flush=True)
for _ in range(3): g.replay()
torch.cuda.synchronize(); print('replays ok', loss_g.item(), flush=True)
