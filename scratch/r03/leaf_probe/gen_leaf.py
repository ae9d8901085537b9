# generates leaf_asm.inc: 16 steps of the register-resident 16x16 leaf as inline asm (v_fmac_f32_dpp row_newbcast)
out = []
for k in range(16):
    n = 16 - k
    # d-step: operands %0..%(n-1) = D[k..15] (rw), %n = rinv (out)
    lines = ["s_nop 1",
             f"v_rsq_f32_dpp %{n}, -%0 row_newbcast:{k} row_mask:0xf bank_mask:0xf bound_ctrl:1",
             "s_nop 0",
             f"v_mul_f32 %0, %0, %{n}",
             "s_nop 1"]
    for c in range(k + 1, 16):
        lines.append(f"v_fmac_f32_dpp %{c-k}, %0, %0 row_newbcast:{c} row_mask:0xf bank_mask:0xf")
    ops_out = ", ".join([f'"+v"(D[{c}])' for c in range(k, 16)] + ['"=&v"(rinv)'])
    out.append('asm volatile("' + '\\n\\t'.join(lines) + '" : ' + ops_out + ');')
    # v-step: %0..%(n-1) = V[k..15] (rw), %n = D[k] (in), %(n+1) = rinv (in)
    lines = [f"v_mul_f32 %0, %0, %{n+1}"]
    for c in range(k + 1, 16):
        lines.append(f"v_fmac_f32_dpp %{c-k}, %{n}, %0 row_newbcast:{c} row_mask:0xf bank_mask:0xf")
    ops_out = ", ".join([f'"+v"(V[{c}])' for c in range(k, 16)])
    out.append('asm volatile("' + '\\n\\t'.join(lines) + '" : ' + ops_out + f' : "v"(D[{k}]), "v"(rinv));')
open("leaf_asm.inc", "w").write("\n".join(out) + "\n")
