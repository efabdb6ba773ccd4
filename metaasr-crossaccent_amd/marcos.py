"""Constants of the reference (src/marcos.py:1-16); the file name keeps the reference's spelling."""
AVAIL_ACCENTS = ['all', 'af', 'au', 'be', 'ca', 'en', 'hk', 'in', 'ir', 'my', 'nz', 'sc', 'ph', 'sg', 'sa', 'us', 'wa']
LOG_DIR = 'testing-logs'
INIT_BEST_ER = 200.0
GRAD_CLIP = 5
IGNORE_ID = -1
SOS_SYMBOL = '<s>'
EOS_SYMBOL = '</s>'
BLANK_SYMBOL = '<blank>'
