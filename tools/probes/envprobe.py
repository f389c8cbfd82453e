import os
print({k: v[:120] for k, v in os.environ.items() if "ROC" in k.upper() or "PRELOAD" in k or "HSA_TOOLS" in k})
