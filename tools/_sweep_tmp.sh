python3 tools/sweep.py --libs default,variants/geo1/libekm_thermo.so,variants/geo2/libekm_thermo.so,variants/geo5/libekm_thermo.so --workloads geopotential --tiles 1 --rounds 6 --steps 5 --out gpurun_out/sweep_geo.json
python3 tools/sweep.py --pmode hybrid --workloads theta,p3,full --tiles 1 --rounds 6 --steps 5 --params hybrid_band_kb=512:2048:4096:16384:65536,lev_per_wg=1:2:4 --out gpurun_out/sweep_hybrid.json
python3 tools/sweep.py --pmode level --workloads theta,p3,full,wetbulb --tiles 1 --rounds 6 --steps 5
python3 tools/sweep.py --pmode field --workloads theta,p3,full,wetbulb,wetbulb_bisect --tiles 1 --rounds 6 --steps 5
