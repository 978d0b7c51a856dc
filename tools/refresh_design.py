import json,re,sys
SHA=sys.argv[1]
src=open('/root/repo/gpurun_out/r04c/sk_phases.txt').read().split("\n")
src=[l for l in src if "amdgpu.ids" not in l]
hdr=open('/root/repo/profiles/r04_mag_sk_phases.txt').read().split("\n")[:2]
hdr[1]=re.sub(r"Kernel sources [0-9a-f]{16}","Kernel sources "+SHA,hdr[1])
open('/root/repo/profiles/r04_mag_sk_phases.txt','w').write("\n".join(hdr+src))
s=open('/root/repo/gpurun_out/r04c/soak.txt').read()
h=re.sub(r"kernel sources [0-9a-f]{16}","kernel sources "+SHA,open('/root/repo/profiles/r04_soak.txt').read().split("\n")[0])
open('/root/repo/profiles/r04_soak.txt','w').write(h+"\n"+s)
b=json.load(open('/root/repo/profiles/r04_bench_lines.json'))
p=json.load(open('/root/repo/profiles/r04_mag_pmc_summary.json'))
assert p['kernel_sha16']==SHA, p['kernel_sha16']
print({k:round(v,3) for k,v in p['derived'].items() if k in ('hbm_traffic_bytes','l2_hit_rate','valu_per_row','salu_per_row')})
for w,l in b.items(): print(w, l['value'], l['roofline']['kernel_ms_avg'], l['roofline']['kernel_sha16'], l['roofline']['traffic'], l['host_api']['rows_per_s'], l['cpu_baseline']['value'], l['roofline']['frac'])
DESIGN='/root/repo/DESIGN.md'
s=open(DESIGN).read()
def fmt(x): return f"{x:,.0f}".replace(","," ")
rows={'MAG-shape':'mag','Reddit-shape':'reddit','Pubmed fixture':'pubmed','Cora fixture':'cora','Amazon2M-shape':'amazon2m'}
out=[]
for line in s.split("\n"):
    for pre,w in rows.items():
        if line.startswith("| "+pre) and line.count("|")==10:
            c=line.split("|"); l=b[w]; bold="**" if w=='mag' else ""
            c[3]=f" {bold}{fmt(l['value'])}{bold} "; c[4]=f" {l['roofline']['kernel_ms_avg']:.2f} "
            c[5]=f" {l['roofline']['achieved']:.1f} ({100*l['roofline']['frac']:.2f} %) "
            c[6]=f" {fmt(l['host_api']['rows_per_s'])} "; c[7]=f" {fmt(l['cpu_baseline']['value'])} "; c[8]=f" {l['detail']['gpu_over_cpu']:.0f}× "
            line="|".join(c)
    out.append(line)
s="\n".join(out)
m=b['mag']
s=re.sub(r"\*\*not met\*\* \([0-9 ]+ in the committed collection","**not met** (%s in the committed collection" % fmt(m['value']),s)
s=re.sub(r"\*\*[0-9.]+ ms per 65 536-row call against [0-9.]+ ms of kernel\*\* \([0-9.]+ ×\)","**%.2f ms per 65 536-row call against %.2f ms of kernel** (%.2f ×)" % (m['host_api']['ms_per_call'], m['roofline']['kernel_ms_avg'], m['host_api']['rows_per_s']/m['value']),s)
s=re.sub(r"host API ≥ 0.97 × device-resident — met \([0-9.]+\)\.","host API ≥ 0.97 × device-resident — met (%.2f)." % (m['host_api']['rows_per_s']/m['value']),s)
s=re.sub(r"\(4\.\d % on the MAG line\)","(%.1f %% on the MAG line)" % (100*m['roofline']['frac']),s)
open(DESIGN,'w').write(s)
