#!/bin/sh
# download_taxondata.sh — NCBI taxonomy dumps for make_metadata.sh (needs network access; same result as CuCLARK's
# script, download_taxondata.sh:31-62):  ./download_taxondata.sh <directory>
# Leaves nodes.dmp, merged.dmp (taxdump.tar.gz) and nucl_accss (nucl_gb + nucl_wgs accession2taxid) in <directory>
# and touches <directory>/../.taxondata.  Earlier dumps in the directory are replaced.
if [ $# -lt 1 ] || [ -z "$1" ]; then
  echo "Usage: $0 <Directory: directory to store taxonomy data> "
  echo "Note: if the chosen directory is not empty, then its content will be erased."
  exit 0
fi
TAXDIR=$1
mkdir -p -m 775 "$TAXDIR"
cd "$TAXDIR" || exit 1
for old in nucl_accss nucl_gb.accession2taxid nucl_wgs.accession2taxid nucl_gb.accession2taxid.gz \
           nucl_wgs.accession2taxid.gz taxdump.tar.gz nodes.dmp merged.dmp names.dmp; do
  [ -f "$old" ] && unlink "$old"
done
echo "Downloading... "
BASE=ftp://ftp.ncbi.nlm.nih.gov/pub/taxonomy
for f in accession2taxid/nucl_gb.accession2taxid.gz accession2taxid/nucl_wgs.accession2taxid.gz taxdump.tar.gz; do
  wget "$BASE/$f"
done
if [ -s nucl_gb.accession2taxid.gz ] && [ -s nucl_wgs.accession2taxid.gz ] && [ -s taxdump.tar.gz ]; then
  echo "Uncompressing files... "
  gunzip nucl_gb.accession2taxid.gz nucl_wgs.accession2taxid.gz
  tar -zxf taxdump.tar.gz
  if [ -s nucl_gb.accession2taxid ] && [ -s nucl_wgs.accession2taxid ] && [ -s nodes.dmp ]; then
    cat nucl_gb.accession2taxid nucl_wgs.accession2taxid > nucl_accss
    touch ../.taxondata
    exit 0
  fi
  echo "Failed to uncompress taxonomy data."
else
  echo "Failed to download taxonomy data!"
fi
exit 1
