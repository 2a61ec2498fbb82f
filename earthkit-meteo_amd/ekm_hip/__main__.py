"""`python -m ekm_hip`: what the library sees."""
import json

from . import device_count, device_info

print(json.dumps([device_info(d) for d in range(device_count())], indent=1))
