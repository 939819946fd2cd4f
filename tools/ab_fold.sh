python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | head -8
for v in 1 0; do echo "fold=$v finetune2: $(SED_LN_FOLD=$v python bench.py --no-cpu-baseline --steps 8 2>/dev/null | tail -1 | cut -c60-100,180-240)"; done
for v in 1 0; do echo "fold=$v pretrain: $(SED_LN_FOLD=$v python bench.py --mode pretrain --no-cpu-baseline --steps 12 2>/dev/null | tail -1 | cut -c90-130,200-260)"; done
for v in 1 0; do echo "fold=$v finetune2: $(SED_LN_FOLD=$v python bench.py --no-cpu-baseline --steps 8 2>/dev/null | tail -1 | cut -c60-100,180-240)"; done
