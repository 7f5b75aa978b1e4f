import sys; sys.path.insert(0,'/root/repo')
import gnuspeech_amd as g
L=g.lib()
print("blocks/CU wide", L.trm_kernel_blocks_per_cu_form(1), "quad(sub=1)", L.trm_kernel_blocks_per_cu_form(2))
