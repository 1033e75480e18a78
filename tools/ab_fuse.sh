# A/B of tuning knobs on one box (developer tool): prints ms/step of the bf16 training epoch per setting
run() { echo "$1"; env $1 timeout -k 10 100 python tools/prof_step.py bf16 16384 2 2>/dev/null | grep epoch | tail -1 || exit 1; }
run "FAVAE_FUSE_MASK=0x3f" && run "FAVAE_FUSE_MASK=0x3b" && run "FAVAE_FUSE_MASK=0x3b FAVAE_PM_TARGET=384" && run "FAVAE_FUSE_MASK=0x3b FAVAE_PM_TARGET=512" && run "FAVAE_FUSE_MASK=0x3b FAVAE_PM_TARGET=768" && run "FAVAE_FUSE_MASK=0x3b FAVAE_NW_BLOCKS=512" && run "FAVAE_FUSE_MASK=0x03"
