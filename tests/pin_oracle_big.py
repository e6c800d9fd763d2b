"""Not a pytest file: a longer pinning run of the oracle against the compiled reference (oracle/_ref/refdump), all six stage dumps,
3 x 30 000 reads on repeat-rich 2.5 Mbp genomes (10-16 regs per read).  ~5 minutes.  python tests/pin_oracle_big.py"""
import sys, os, subprocess, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import ref_binary
from tools import oracle, refio, synth
exe, dump = ref_binary(), ref_binary("refdump")
for seed in (1001, 1002, 1003):
    d=tempfile.mkdtemp(prefix="pin")
    rng=np.random.default_rng(seed)
    names, ctg, alts = synth.make_genome(seed, [1500000, 700000, 200000, 50000], alt_contigs=3, alt_len=20000, n_repeat_families=40, repeat_len=(100, 6000),
                                         copies=(3, 300), divergence=(0.0, 0.15), n_gaps=8, gap_len=(50, 3000))
    fa=os.path.join(d,"g.fa"); synth.write_fasta(fa,names,ctg); synth.write_alt(fa+".alt",alts)
    subprocess.check_call([exe,"index",fa],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL)
    L=int(rng.choice([100,150,250]))
    reads = synth.make_reads_se(seed+1, ctg, 30000, L=L, sub_rate=float(rng.choice([0.005,0.02,0.05])), indel_frac=float(rng.choice([0.05,0.3])), random_frac=0.01)
    rt=os.path.join(d,"reads.txt")
    with open(rt,"w") as f:
        for r in reads: f.write("".join("ACGTN"[c] for c in r)+"\n")
    enc,off,ln=refio.pack_reads(list(reads))
    out=os.path.join(d,"dump")
    t0=time.time(); subprocess.check_call([dump,fa,rt,out],stderr=subprocess.DEVNULL); t1=time.time()
    dd=refio.read_dump(out)
    ix=oracle.Index(fa)
    exp=ix.run(enc,off,ln); t2=time.time()
    ix.close()
    bad=[t for t in ("SMEM","SACOORD","CHN1","SEED1","REGRAW","REGPRG") if dd[t].tobytes()!=exp[t].tobytes()]
    print("seed",seed,"L",L,"reads",len(reads),"regs",len(exp["REGPRG"]),"max regs/read", np.bincount(exp["REGPRG"]["read"]).max(), "ref %.0fs oracle %.0fs"%(t1-t0,t2-t1), "DIFF "+str(bad) if bad else "all stages identical", flush=True)
