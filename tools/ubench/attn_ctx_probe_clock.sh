for v in "-DMDT_ABL_CTX_CLOCK" "-DMDT_ABL_CTX_CLOCK -DCTXP_SPLIT" "-DMDT_ABL_CTX_CLOCK -DMDT_ABL_CTX_NODMA" "-DMDT_ABL_CTX_CLOCK -DMDT_ABL_CTX_NODMA -DCTXP_SPLIT"; do
  echo "== variant: $v"
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMDT_TUNING $v -Imoleculediffusiontransformer_amd/csrc -Iinclude tools/ubench/attn_ctx_probe.hip -o /tmp/attn_ctx_probe 2>/dev/null && /tmp/attn_ctx_probe
done
