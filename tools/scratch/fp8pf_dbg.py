import torch
from revisionllm_amd import engine
from revisionllm_amd.utils import synth
shape = synth.LlamaShape(layers=2, vocab=2048)
eng = engine.Engine(shape, adapter_text=False, device="cuda:0")
eng.init_synthetic(seed=1, llm=True, clip=False, fp8_decode=True, fp8_prefill=True, cond=synth.CONDITIONED)
print("defaults", eng.get_option("fp8_prefill"), eng.get_option("fp8_decode"))
D=4096; B,S,P0=7,150,32
G=4
h = (torch.randn(G*(P0+B*(S-P0)), D, device="cuda:0")*0.5)
pool, Smax = eng.new_kv_pool(70, 192)
def run():
    return eng.llm_prefill_pool_groups(h.clone(), G, B, P0, pool, 70, [7*i for i in range(G)], Smax).clone()
a = run()
eng.set_option("fp8_prefill", 0); b = run()
eng.set_option("fp8_prefill", 1); c = run()
print("default vs off", float((a-b).abs().max()/b.abs().max()), "on vs off", float((c-b).abs().max()/b.abs().max()), "rows", h.shape[0])
